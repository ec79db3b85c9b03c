"""Host-visible latency of the per-frame entry points a drop-in caller uses besides predict / update, at a small map:
python tools/api_latency.py [N]"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R]
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = pkg.kinect_config()
px0, z = synthetic.measurement_stream(cfg, N, 20, sigma_px=0.5)
f = pkg.VSlamFilter(cfg, capacity_features=N + 8)
f.setDt(1 / 30.0)
for (u, v) in px0:
    assert f.addFeature((u, v)) == 1
idx = np.arange(N, dtype=np.int32)
for k in range(5):
    f.predict(); f.update(z[k].reshape(-1).astype(np.float32), idx)
f.synchronize()

def timeit(name, fn, reps=200, sync=True):
    fn(); f.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    if sync: f.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) / reps * 1e6:8.1f} us")

timeit("getState (camera, 13/14 scalars)", lambda: f.getState())
timeit("getSigma (camera block)", lambda: f.getSigma())
timeit("Covariance_Parameter", lambda: f.Covariance_Parameter())
timeit("featureXYZ(3)", lambda: f.featureXYZ(3))
timeit("searchEllipses", lambda: (f.predict(), f.searchEllipses()), reps=100)
timeit("getPointsFeatures", lambda: f.getPointsFeatures())
timeit("getPointsTable", lambda: f.getPointsTable())
timeit("convert2XYZ_ifLinearAll (none linear)", lambda: f.convert2XYZ_ifLinearAll(), reps=50)
def add_remove():
    assert f.addFeature((200.0 + np.random.rand() * 50, 150.0)) == 1
    f.removeFeature(f.numOfFeatures() - 1)
timeit("addFeature + removeFeature", add_remove, reps=50)
