import sys, os, time
import numpy as np
sys.path.insert(0, '/root/repo')
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import synthetic
N = 4000
cfg = pkg.kinect_config()
px0, z = synthetic.measurement_stream(cfg, N, 2, sigma_px=0.5)
f = pkg.VSlamFilter(cfg, capacity_features=N + 64)
f.setDt(1 / 30.0)
for (u, v) in px0:
    f.addFeature((u, v))
f.synchronize()
rng = np.random.default_rng(3)
for it in range(5):
    n = f.numOfFeatures()
    drop = sorted(rng.choice(n, size=40, replace=False).tolist())
    f.synchronize(); t0 = time.perf_counter()
    f.removeFeatures(drop)
    f.synchronize(); dt = time.perf_counter() - t0
    nn = f.stateDim()
    print(f"removal {it}: {dt * 1e3:.3f} ms  ({2 * nn * nn * 4 / dt / 1e12:.2f} TB/s of 2 n^2 s)")
