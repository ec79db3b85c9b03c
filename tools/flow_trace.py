"""Per-task timeline of the persistent dataflow launch (EKF_FLOW_TRACE=1): utilisation, waits by task type."""
import os, sys, ctypes as C
import numpy as np
os.environ["EKF_FLOW_TRACE"] = "1"
os.environ["EKF_FLOW"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import synthetic
cfg = pkg.kinect_config()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
px0, z = synthetic.measurement_stream(cfg, N, 6, sigma_px=0.5)
f = pkg.VSlamFilter(cfg, capacity_features=N)
f.setDt(1 / 30.0)
for (u, v) in px0:
    f.addFeature((u, v))
idx = np.arange(N, dtype=np.int32)
for k in range(5):
    f.predict(); f.update(z[k].reshape(-1), idx)
f.synchronize()
nt = C.c_int()
f._lib.ekf_debug_flow_trace(f._h, None, 0, C.byref(nt))
buf = np.zeros((nt.value, 4), np.uint64)
f._lib.ekf_debug_flow_trace(f._h, buf.ctypes.data_as(C.c_void_p), nt.value, C.byref(nt))
t0 = buf[:, 0].min()
fetch, ready, done = [(buf[:, i] - t0).astype(np.float64) / 100.0 for i in range(3)]     # microseconds
wg = (buf[:, 3] & 0xffffffff).astype(int)
typ = ((buf[:, 3] >> 32) & 0xff).astype(int)
grid = ((buf[:, 3] >> 40)).astype(int)
names = ["solve", "wupdate", "downdate"]
print(f"tasks {nt.value}  span {done.max():.1f} us  (first fetch -> last store)")
for t in range(3):
    s = typ == t
    print(f"{names[t]:9s} n {s.sum():5d}  wait mean {np.mean(ready[s]-fetch[s]):7.2f} us  max {np.max(ready[s]-fetch[s]):7.1f}   run mean {np.mean(done[s]-ready[s]):7.2f} us"
          f"   first ready {ready[s].min():7.1f}  last done {done[s].max():7.1f}")
# utilisation over time: number of workgroups computing, 25 us bins
T = done.max()
bins = np.arange(0, T + 25, 25)
busy = np.zeros(len(bins) - 1)
wait = np.zeros(len(bins) - 1)
for a, b, c in zip(fetch, ready, done):
    for arr, lo, hi in ((wait, a, b), (busy, b, c)):
        i0, i1 = int(lo // 25), int(min(hi, T) // 25)
        for i in range(i0, min(i1 + 1, len(arr))):
            arr[i] += (min(hi, bins[i + 1]) - max(lo, bins[i])) / 25.0
print("t[us]  computing  waiting   (workgroups, of", np.unique(grid), ")")
for i in range(len(busy)):
    print(f"{bins[i]:6.0f} {busy[i]:9.1f} {wait[i]:9.1f}")
