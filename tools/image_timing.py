"""Timing of the image-side entry points at N = 1000 (frame upload, predict with blur, NCC search)."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "oracle")]
from __graft_entry__ import load_package
pkg = load_package()
import image_oracle as io_
from ekf_monoslam_amd import synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cfg = dict(pkg.kinect_config()); cfg.update(kernel_size=2, T_camera=0.5)
px0, z = synthetic.measurement_stream(cfg, N, 2, sigma_px=0.5)
frame = io_.random_texture(240, 320, seed=5)
f = pkg.VSlamFilter(cfg, capacity_features=N)
f.setDt(1 / 30.0)
f.setFrame(frame)
for (u, v) in px0:
    assert f.addFeature((u, v)) == 1
mu = f.getFullState(); mu[10:13] = (0.3, 1.2, -0.2); f.setFullState(mu)
f.set_option(2, 2)
for it in range(3):
    f.profile_reset()
    f.synchronize(); t0 = time.perf_counter()
    f.setFrame(frame); t1 = time.perf_counter()
    f.predict(); f.synchronize(); t2 = time.perf_counter()
    zz, found, score = f.findMatches(); t3 = time.perf_counter()
    prof = f.profile()
print("setFrame %.1f us   predict(+blur) %.1f us   findMatches (incl. D2H of z/found/score) %.1f us   found %d / %d" %
      (1e6 * (t1 - t0), 1e6 * (t2 - t1), 1e6 * (t3 - t2), int(found.sum()), N))
print({k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items()})
