"""A/B of the persistent dataflow launch (EKF_FLOW=1, default) against the launch-per-phase path (EKF_FLOW=0):
bit-identity of mu / Sigma after two frames at N = 1000 and N = 640, then step time at N = 1000."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from __graft_entry__ import load_package
pkg = load_package()
import torch
from ekf_monoslam_amd import synthetic
cfg = pkg.kinect_config()
def build(N, px0, flow):
    os.environ["EKF_FLOW"] = str(flow)
    f = pkg.VSlamFilter(cfg, capacity_features=N)
    f.setDt(1 / 30.0)
    for (u, v) in px0:
        f.addFeature((u, v))
    return f
for N in (640, 1000):
    px0, z = synthetic.measurement_stream(cfg, N, 3, sigma_px=0.5)
    idx = np.arange(N, dtype=np.int32)
    outs = []
    for flow in (0, 1):
        f = build(N, px0, flow)
        for k in range(2):
            f.predict(); f.update(z[k].reshape(-1), idx)
        f.synchronize()
        outs.append((f.getFullState(), f.getFullSigma()))
    print(N, "bit-identical mu", np.array_equal(outs[0][0], outs[1][0]), "Sigma", np.array_equal(outs[0][1], outs[1][1]),
          "max |dSigma|", np.abs(outs[0][1] - outs[1][1]).max(), flush=True)
N = 1000
frames = 130
px0, z = synthetic.measurement_stream(cfg, N, frames, sigma_px=0.5)
d_z = torch.from_numpy(z.reshape(frames, -1)).cuda().contiguous()
d_idx = torch.arange(N, dtype=torch.int32, device="cuda")
for flow in (0, 1, 0, 1):
    f = build(N, px0, flow)
    for k in range(10):
        f.predict(); f.update_device(d_z.data_ptr() + k * 8 * N, d_idx.data_ptr(), N)
    f.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(10, 110):
        f.predict(); f.update_device(d_z.data_ptr() + k * 8 * N, d_idx.data_ptr(), N)
    f.synchronize()
    dt = (time.perf_counter() - t0) / 100
    print("EKF_FLOW", flow, "ms/step %.4f" % (dt * 1e3), "updates/s %.1f" % (1 / dt), flush=True)
