"""fp32 accuracy of the launch structures against the fp64 filter on the same stream (4 frames, every feature measured):
default (chunked factorisation), serial (EKF_OPT_PIPELINE = 0: one chunk, explicit inverse of the whole factor) and the VALU
tiles.  Measured: default and VALU 1.1-1.6e-5 of max|Sigma| at N = 1200 / 2000, serial 1.2-1.4e-4."""
import os, sys
import numpy as np
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import synthetic
cfg = pkg.kinect_config()
for N in ([int(a) for a in sys.argv[1:]] or [1200, 2000]):
    px0, z = synthetic.measurement_stream(cfg, N, 4, sigma_px=0.5)
    idx = np.arange(N, dtype=np.int32)
    res = {}
    for name, dt, opts in (("f64", np.float64, ()), ("default", np.float32, ()), ("serial", np.float32, ((3, 0),)), ("valu", np.float32, ((1, 0),))):
        f = pkg.VSlamFilter(cfg, capacity_features=N, dtype=dt)
        f.setDt(1 / 30.0)
        for (u, v) in px0:
            assert f.addFeature((u, v)) == 1
        for k, v in opts:
            f.set_option(k, v)
        for k in range(4):
            f.predict()
            f.update(z[k].reshape(-1).astype(dt), idx)
        f.synchronize()
        res[name] = (f.getFullState().astype(np.float64), f.getFullSigma().astype(np.float64))
        f.close()
    mu0, S0 = res["f64"]
    for name in ("default", "serial", "valu"):
        mu, S = res[name]
        print(f"N={N} {name:8s} vs fp64 filter: rel|mu| {np.abs(mu - mu0).max() / np.abs(mu0).max():.2e}  rel|Sigma| {np.abs(S - S0).max() / np.abs(S0).max():.2e}", flush=True)
