"""Inside the persistent chain kernel (csrc/ekf_chain.hpp): per-task time stamps of ONE update at N features, from the
device-side trace (EKF_CHAIN_TRACE=1, ekf_peek_workspace which = 2).  Prints the critical workgroup's sequence, per block
step the span of the bulk work, and the waiting / computing / publishing totals.
    python tools/chain_trace.py [N] [warm-up updates]"""
import os, sys
import numpy as np
os.environ["EKF_CHAIN_TRACE"] = "1"
os.environ["EKF_CHAIN_PERSISTENT"] = "1"          # (opt-in since the measurement this tool made: DESIGN 5)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = pkg.kinect_config()
px0, z = synthetic.measurement_stream(cfg, N, warm + 1, sigma_px=0.5)
f = pkg.VSlamFilter(cfg, capacity_features=N)
f.setDt(1 / 30.0)
for (u, v) in px0:
    f.addFeature((u, v))
idx = np.arange(N, dtype=np.int32)
for k in range(warm + 1):
    f.predict()
    f.update(z[k].reshape(-1), idx)
f.synchronize()
hdr = f.peekWorkspace(2, -1, 0, 1, 8).view(np.uint32)
nrec = int(hdr[0, 0])
rec = f.peekWorkspace(2, 0, 0, min(nrec, 65536), 8).view(np.uint32).astype(np.int64)
print(f"N = {N}: chunk plan {f.chunkPlan()}, {nrec} tasks traced, launches {f.launch_counts()['chain_persistent']}")
typ = rec[:, 0] & 0xff
wg = (rec[:, 0] >> 8) & 0xffff
crit = (rec[:, 0] >> 24) & 1
t = rec[:, 4:8].copy()
t0 = t[:, 0].min()
t = (t - t0) * 0.01                                          # us (100 MHz)
names = {0: "D", 1: "P", 2: "T"}
for cde in range(16):
    names[16 + cde] = f"m{cde}"
for q in range(4):
    names[4 + q] = "Q"
order = np.argsort(t[:, 0])
print("critical workgroup (us: draw, deps met, computed, published | wait, compute, publish):")
for i in order:
    if typ[i] >= 16:
        print(f"      mark {int(typ[i]) - 16} (step {rec[i, 1]:2d})  {t[i, 0]:8.1f}")
    elif crit[i]:
        print(f"  {names[int(typ[i])]}({rec[i, 1]:2d}; {rec[i, 2]:2d}, {rec[i, 3]:2d})  {t[i, 0]:8.1f} {t[i, 1]:8.1f} {t[i, 2]:8.1f} {t[i, 3]:8.1f} | "
              f"{t[i, 1] - t[i, 0]:6.1f} {t[i, 2] - t[i, 1]:6.1f} {t[i, 3] - t[i, 2]:6.1f}")
marks = (typ >= 16)
crit = crit & ~marks
print("bulk work per block step (first draw .. last publish, tasks, sum of wait / compute / publish over workgroups, us):")
for j in sorted(set(rec[:, 1].tolist())):
    for ty in (1, 2, 4):
        sel = (rec[:, 1] == j) & ((typ == ty) if ty < 4 else ((typ >= 4) & (typ < 8))) & (crit == 0)
        if sel.any():
            print(f"  step {j:2d} {names[ty]}: {t[sel, 0].min():8.1f} .. {t[sel, 3].max():8.1f}  {int(sel.sum()):4d} tasks  wait {np.sum(t[sel, 1] - t[sel, 0]):8.1f}  "
                  f"compute {np.sum(t[sel, 2] - t[sel, 1]):8.1f} (mean {np.mean(t[sel, 2] - t[sel, 1]):5.2f})  publish {np.sum(t[sel, 3] - t[sel, 2]):7.1f}  on {len(set(wg[sel].tolist()))} workgroups")
print(f"total span {t[:, 3].max():.1f} us")
