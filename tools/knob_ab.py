"""Step time at N = 1000 for a list of environment settings: python tools/knob_ab.py "EKF_SIDE2=1" "EKF_SIDE2=1 EKF_CHUNKS=3,9,16" ...
("" = defaults).  Each setting: build, 10 warm-up steps, 150 timed steps with resident inputs; also checks mu against the default."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from __graft_entry__ import load_package
pkg = load_package()
import torch
from ekf_monoslam_amd import synthetic
cfg = pkg.kinect_config()
N = int(os.environ.get("KNOB_N", "1000"))
frames = int(os.environ.get("KNOB_FRAMES", "170"))      # 10 warm-up + (frames - 20) timed steps
px0, z = synthetic.measurement_stream(cfg, N, frames, sigma_px=0.5)
d_z = torch.from_numpy(z.reshape(frames, -1)).cuda().contiguous()
d_idx = torch.arange(N, dtype=torch.int32, device="cuda")
base = None
settings = sys.argv[1:] or [""]
keys = set()
for st in settings:
    for kv in st.split():
        keys.add(kv.split("=")[0])
for rep in range(2):
    for st in settings:
        for k in keys:
            os.environ.pop(k, None)
        for kv in st.split():
            k, v = kv.split("=")
            os.environ[k] = v
        f = pkg.VSlamFilter(cfg, capacity_features=N)
        f.setDt(1 / 30.0)
        for (u, v) in px0:
            f.addFeature((u, v))
        for k in range(10):
            f.predict(); f.update_device(d_z.data_ptr() + k * 8 * N, d_idx.data_ptr(), N)
        f.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(10, frames - 10):
            f.predict(); f.update_device(d_z.data_ptr() + k * 8 * N, d_idx.data_ptr(), N)
        f.synchronize()
        dt = (time.perf_counter() - t0) / (frames - 20)
        mu = f.getFullState()
        if base is None:
            base = mu
        print(f"[{st or 'defaults':40s}] {dt * 1e3:.4f} ms/step  {1 / dt:7.1f} updates/s   max|mu - mu_default| {np.abs(mu - base).max():.2e}", flush=True)
        f.close()
