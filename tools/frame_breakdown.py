"""Per-kernel HIP-event breakdown of one EKF step (EKF_OPT_PROFILE = 2) + run sanity details."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
import torch
from ekf_monoslam_amd import synthetic
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = pkg.kinect_config()
px0, z = synthetic.measurement_stream(cfg, N, steps + 3, sigma_px=bench.SIGMA_Z_PX)
flt = bench.build_filter(pkg, cfg, N, px0)
pipe = int(sys.argv[3]) if len(sys.argv) > 3 else -1
M = int(sys.argv[4]) if len(sys.argv) > 4 else N
flt.set_option(3, pipe)
dev = torch.device("cuda", 0)
d_z = torch.from_numpy(z.reshape(z.shape[0], -1)).to(dev).contiguous()
d_idx = torch.arange(M, dtype=torch.int32, device=dev)
d_z = d_z[:, :2 * M].contiguous()
bpf = 2 * M * 4
bench.run_steps(flt, d_z, d_idx, M, 0, 3, bpf)
flt.synchronize()
flt.set_option(2, 2)
flt.profile_reset()
t0 = time.perf_counter()
bench.run_steps(flt, d_z, d_idx, M, 3, steps, bpf)
flt.synchronize()
t1 = time.perf_counter()
prof = flt.profile()
tot = 0.0
for k, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:22s} total {ms:9.3f} ms  launches {cnt:6d}  avg {ms/cnt*1e3:9.2f} us  per-step {ms/steps:8.4f} ms")
    tot += ms
print(f"sum of kernels per step {tot/steps:.4f} ms ; wall per step (with events) {(t1-t0)/steps*1e3:.4f} ms")
mu = flt.getFullState()
flt.predict()
h, vis, rem, S2 = flt.predictions()
print("finite", np.all(np.isfinite(mu)), "|q|", np.linalg.norm(mu[3:7]), "visible", int(vis.sum()), "/", N, "remove", int(rem.sum()))
r_true, q_true = synthetic.trajectory((steps + 3) / 30.0)
print("r est", mu[0:3], "true", r_true)
print("q est", mu[3:7], "true", q_true)
print("rho  min/med/max", mu[14:][5::6].min(), np.median(mu[14:][5::6]), mu[14:][5::6].max())
zz = z[steps + 2]
print("innovation rms px (next frame pred vs last z)", np.sqrt(np.mean((h[:M] - zz[:M]) ** 2)))
