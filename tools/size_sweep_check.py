"""Consistency of the update across launch structures at sizes the parity tests do not reach: the default pipelined path
(chunked factorisation, half tiles, queued launches) against the serial path (EKF_OPT_PIPELINE = 0) and the plain VALU
tiles (EKF_OPT_USE_MFMA = 0), 4 frames each, relative differences of mu and Sigma.  usage: size_sweep_check.py N [N ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import synthetic
cfg = pkg.kinect_config()
for N in [int(a) for a in sys.argv[1:]] or [1500]:
    px0, z = synthetic.measurement_stream(cfg, N, 4, sigma_px=0.5)
    idx = np.arange(N, dtype=np.int32)
    res = {}
    for name, opts in (("default", ()), ("serial", ((3, 0),)), ("valu", ((1, 0),))):
        f = pkg.VSlamFilter(cfg, capacity_features=N)
        f.setDt(1 / 30.0)
        for (u, v) in px0:
            assert f.addFeature((u, v)) == 1
        for k, v in opts:
            f.set_option(k, v)
        for k in range(4):
            f.predict()
            f.update(z[k].reshape(-1), idx)
        f.synchronize()
        pad, asym, big = f.checkInvariants()
        res[name] = (f.getFullState(), f.getFullSigma(), pad, asym)
        f.close()
    mu0, S0 = res["default"][0], res["default"][1]
    for name in ("serial", "valu"):
        mu, S = res[name][0], res[name][1]
        print(f"N={N:5d} default vs {name:7s}: rel|mu| {np.abs(mu - mu0).max() / np.abs(mu0).max():.2e}  rel|Sigma| {np.abs(S - S0).max() / np.abs(S0).max():.2e}"
              f"  invariants pad {res[name][2]:.1e} asym {res[name][3]:.1e} (default: {res['default'][2]:.1e} {res['default'][3]:.1e})", flush=True)
