"""k_propagate_streaming (2 n^2 s = 289 MB at N = 1000) timed by the library's HIP events, un-profiled: launches back to back,
and launches with the GPU left idle for a few milliseconds in front of each one.  (under rocprofv3
--kernel-trace, where every dispatch waits for the completion signal of the one before it and a step stretches from 1.1
to 2.5 ms, the kernel takes 77-125 us instead of 42-49.)  usage: python3 tools/propagate_idle_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import synthetic
cfg = pkg.kinect_config()
N = 1000
px0, z = synthetic.measurement_stream(cfg, N, 2, sigma_px=0.5)
f = pkg.VSlamFilter(cfg, capacity_features=N)
f.setDt(1 / 30.0)
for (u, v) in px0:
    f.addFeature((u, v))
f.set_option(0, 1)                      # EKF_OPT_PROPAGATE_STREAMING
n = f.stateDim()
for idle_ms in (0.0, 0.2, 1.0, 3.0, 10.0):
    f.set_option(2, 1)
    f.profile_reset()
    for k in range(60):
        f.predict()
        if idle_ms:
            f.synchronize()
            time.sleep(idle_ms * 1e-3)
    f.synchronize()
    ms, cnt = f.profile()["propagate_streaming"]
    t = ms / cnt * 1e-3
    print(f"idle {idle_ms:5.1f} ms in front of every launch: {cnt} launches, mean {t * 1e6:7.1f} us = {2.0 * n * n * 4 / t / 1e12:5.2f} TB/s", flush=True)
    f.set_option(2, 0)
f.close()
