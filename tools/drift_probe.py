"""HIP-only positivity drift probe: min eigenvalue of Sigma every `every` frames (N = 200 stream, all measured)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import synthetic
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 500
dt = np.float64 if (len(sys.argv) > 4 and sys.argv[4] == "f64") else np.float32
cfg = pkg.kinect_config()
px0, zs = synthetic.measurement_stream(cfg, N, frames, sigma_px=bench.SIGMA_Z_PX)
g = pkg.VSlamFilter(cfg, capacity_features=N, dtype=dt)
g.setDt(1 / 30.0)
if os.environ.get("EKF_NOISE"):
    g.set_option(5, int(os.environ["EKF_NOISE"]))      # EKF_OPT_FEATURE_NOISE, units of 1e-12 per predict
if os.environ.get("EKF_MFMA"):
    g.set_option(1, int(os.environ["EKF_MFMA"]))       # EKF_OPT_USE_MFMA: 0 = plain VALU tiles
for (u, v) in px0:
    g.addFeature((u, v))
idx = np.arange(N, dtype=np.int32)
for k in range(frames):
    g.predict()
    if k % every == 0:
        P = g.getFullSigma().astype(np.float64)
        w, U = np.linalg.eigh(P)
        u = U[:, 0]
        top = np.argsort(-np.abs(u))[:6]
        cam = float(np.sum(u[:14] ** 2))
        desc = " ".join(f"{i}:{u[i]:+.2f}" for i in top)
        kinds = {}
        for i in top:
            kinds[(i - 14) % 6 if i >= 14 else -1] = kinds.get((i - 14) % 6 if i >= 14 else -1, 0) + 1
        print(f"{k:6d} min eig {w[0]: .3e}  #neg<-1e-9 {(w < -1e-9).sum():4d}  max {w[-1]:.3e}  |u_cam|^2 {cam:.3f}  top {desc}  comp-kinds {kinds}", flush=True)
    try:
        g.update(zs[k].reshape(-1).astype(dt), idx); g.synchronize()
    except Exception as e:
        print("stopped at", k, e); break
