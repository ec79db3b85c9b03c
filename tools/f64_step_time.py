"""Step time of the fp64 filter at N = M features (EKF_OPT_USE_MFMA on / off)."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R]
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = 40
cfg = pkg.kinect_config()
px0, z = synthetic.measurement_stream(cfg, N, steps + 5, sigma_px=0.5)
idx = np.arange(N, dtype=np.int32)
for mfma in (1, 0):
    f = pkg.VSlamFilter(cfg, capacity_features=N, dtype=np.float64)
    f.set_option(1, mfma)
    f.setDt(1 / 30.0)
    for (u, v) in px0:
        assert f.addFeature((u, v)) == 1
    zz = z.astype(np.float64)
    for k in range(5):
        f.predict(); f.update(zz[k].reshape(-1), idx)
    f.synchronize(); t0 = time.perf_counter()
    for k in range(5, 5 + steps):
        f.predict(); f.update(zz[k].reshape(-1), idx)
    f.synchronize(); t1 = time.perf_counter()
    print(f"fp64 N={N} mfma={mfma}: {1e3 * (t1 - t0) / steps:.3f} ms/step  ({steps / (t1 - t0):.1f} updates/s, host-buffer update)")
    f.close()

f = pkg.VSlamFilter(cfg, capacity_features=N, dtype=np.float64)
f.setDt(1 / 30.0)
for (u, v) in px0:
    assert f.addFeature((u, v)) == 1
zz = z.astype(np.float64)
for k in range(3):
    f.predict(); f.update(zz[k].reshape(-1), idx)
f.synchronize(); f.set_option(2, 2); f.profile_reset()
for k in range(3, 13):
    f.predict(); f.update(zz[k].reshape(-1), idx)
f.synchronize()
for name, (ms, cnt) in sorted(f.profile().items(), key=lambda kv: -kv[1][0]):
    print(f"  {name:20s} {ms / 10:8.4f} ms/step  {cnt // 10:4d} launches/step  avg {1e3 * ms / max(cnt, 1):8.1f} us")
