"""Per-kernel HIP-event time of the step at N features (EKF_OPT_PROFILE = 2: events around every launch; the step itself is
slower under it): python tools/kernel_ms.py [N] [steps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from __graft_entry__ import load_package
pkg = load_package()
import torch
from ekf_monoslam_amd import synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = pkg.kinect_config()
px0, z = synthetic.measurement_stream(cfg, N, steps + 10, sigma_px=0.5)
d_z = torch.from_numpy(z.reshape(z.shape[0], -1)).cuda().contiguous()
d_idx = torch.arange(N, dtype=torch.int32, device="cuda")
f = pkg.VSlamFilter(cfg, capacity_features=N)
f.setDt(1 / 30.0)
for (u, v) in px0:
    f.addFeature((u, v))
for k in range(10):
    f.predict(); f.update_device(d_z.data_ptr() + k * 8 * N, d_idx.data_ptr(), N)
f.synchronize()
f.set_option(2, 2)
f.profile_reset()
for k in range(10, 10 + steps):
    f.predict(); f.update_device(d_z.data_ptr() + k * 8 * N, d_idx.data_ptr(), N)
f.synchronize()
n = f.stateDim()
print(f"N = {N}, n = {n}: mean HIP-event time per launch over {steps} steps (launches per step in brackets)")
for name, (ms, cnt) in sorted(f.profile().items(), key=lambda kv: -kv[1][0]):
    print(f"  {name:22s} {1e3 * ms / cnt:8.2f} us  [{cnt / steps:5.2f}]   {1e3 * ms / steps:8.1f} us per step")
