"""Config-5-shaped sanity run: N features, dynamic remove + add every `period` frames.
Prints timing of the resize operations and of the steps; checks the state stays finite."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
import torch
from ekf_monoslam_amd import synthetic
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
period = int(sys.argv[3]) if len(sys.argv) > 3 else 25
cfg = pkg.kinect_config()
rng = np.random.default_rng(1236)
px0, z = synthetic.measurement_stream(cfg, N, frames, sigma_px=bench.SIGMA_Z_PX)
t0 = time.perf_counter()
flt = bench.build_filter(pkg, cfg, N, px0)
print(f"N={N} n={flt.stateDim()} build (N sequential addFeature) {time.perf_counter()-t0:.2f} s")
dev = torch.device("cuda", 0)
alive = np.arange(N)                      # stream column of each live feature
d_z_all = torch.from_numpy(z).to(dev)
t_steps = []
for f in range(frames):
    if f and f % period == 0:
        k = max(1, N // 100)
        idx = np.sort(rng.choice(len(alive), size=k, replace=False))
        flt.synchronize(); t0 = time.perf_counter()
        flt.removeFeatures(idx)
        flt.synchronize(); t1 = time.perf_counter()
        removed_cols = alive[idx]
        alive = np.delete(alive, idx)
        for c in removed_cols:           # re-add the same world points at their current pixel
            assert flt.addFeature(z[f - 1, c]) == 1
        flt.synchronize(); t2 = time.perf_counter()
        alive = np.concatenate([alive, removed_cols])
        print(f"frame {f}: removed {k} features in {1e3*(t1-t0):.2f} ms (one compaction pass), re-added in {1e3*(t2-t1):.2f} ms; N={flt.numOfFeatures()}")
    zz = d_z_all[f][torch.from_numpy(alive).to(dev)].contiguous().reshape(-1)
    idxs = torch.arange(len(alive), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    flt.predict()
    flt.update_device(zz.data_ptr(), idxs.data_ptr(), len(alive), False)
    flt.synchronize(); t_steps.append(time.perf_counter() - t0)
mu = flt.getFullState()
h, vis, rem, S2 = (flt.predict(), flt.predictions())[1]
print(f"steps: median {1e3*np.median(t_steps):.2f} ms  ({1/np.median(t_steps):.1f} updates/s); finite={np.all(np.isfinite(mu))} |q|={np.linalg.norm(mu[3:7]):.6f} visible={int(vis.sum())}/{len(vis)} rho<=0: {int(rem.sum())}")
