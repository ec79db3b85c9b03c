import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from __graft_entry__ import load_package
pkg = load_package()
import torch
from ekf_monoslam_amd import synthetic
N = 200; frames = 620
cfg = pkg.kinect_config()
px0, z = synthetic.measurement_stream(cfg, N, frames, sigma_px=0.5)
d_z = torch.from_numpy(z.reshape(frames, -1)).cuda().contiguous()
d_idx = torch.arange(N, dtype=torch.int32, device="cuda")
for rep in range(2):
    for nch in (0, 2, 3, 4):
        f = pkg.VSlamFilter(cfg, capacity_features=N)
        f.setDt(1 / 30.0)
        for (u, v) in px0:
            f.addFeature((u, v))
        if nch:
            f.set_option(3, nch)
        for k in range(10):
            f.predict(); f.update_device(d_z.data_ptr() + k * 8 * N, d_idx.data_ptr(), N)
        f.synchronize()
        t0 = time.perf_counter()
        for k in range(10, frames - 10):
            f.predict(); f.update_device(d_z.data_ptr() + k * 8 * N, d_idx.data_ptr(), N)
        f.synchronize()
        dt = (time.perf_counter() - t0) / (frames - 20)
        print(f"option 3 = {nch}: {dt * 1e3:.4f} ms/step  plan {f.chunkPlan()[1]}", flush=True)
        f.close()
