"""Long free run at N features: watch the covariance while every feature is measured every frame."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R]
from __graft_entry__ import load_package
pkg = load_package()
import torch
from ekf_monoslam_amd import synthetic
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
pipe = int(sys.argv[3]) if len(sys.argv) > 3 else -1
WATCH = int(sys.argv[4]) if len(sys.argv) > 4 else 0
DT = np.float64 if (len(sys.argv) > 5 and sys.argv[5] == 'f64') else np.float32
cfg = pkg.kinect_config()
px0, z = synthetic.measurement_stream(cfg, N, steps, sigma_px=bench.SIGMA_Z_PX)
flt = pkg.VSlamFilter(cfg, capacity_features=N, dtype=DT)
flt.setDt(1.0 / 30.0)
for (u, v) in px0:
    assert flt.addFeature((u, v)) == 1
flt.set_option(3, pipe)
idx = np.arange(N, dtype=np.int32)
for k in range(steps):
    flt.predict()
    if k % 50 == 0 or k > steps - 3 or (WATCH and abs(k - WATCH) <= 3):
        h, vis, rem, S2 = flt.predictions()
        nb_ = min(600, flt.stateDim() - 14)
        d = np.diag(flt.getSigmaBlock(14, 14, nb_, nb_))
        cam = np.diag(flt.getSigma())
        inn = np.sqrt(np.mean((h - z[k].reshape(-1, 2)) ** 2))
        zz = z[k].reshape(-1, 2)
        print(f"   vis {int(vis.sum())} rem {int(rem.sum())}  h range u [{h[:,0].min():.1f},{h[:,0].max():.1f}] v [{h[:,1].min():.1f},{h[:,1].max():.1f}]  z range u [{zz[:,0].min():.1f},{zz[:,0].max():.1f}] v [{zz[:,1].min():.1f},{zz[:,1].max():.1f}]  max|h-z| {np.abs(h-zz).max():.2f}  S2 max {S2.max():.1f} finite {np.isfinite(S2).all()}")
        print(f"step {k:5d}  S2 diag min {S2[:, 0, 0].min():.4f} {S2[:, 1, 1].min():.4f}  Sigma feat diag min {d.min():.3e} max {d.max():.3e}  cam diag min {cam.min():.3e}  rms innovation {inn:.3f} px", flush=True)
    try:
        flt.update(z[k].reshape(-1).astype(DT), idx)
        flt.synchronize()
    except Exception as e:
        print("FAILED at step", k, e)
        h, vis, rem, S2 = flt.predictions()
        break
