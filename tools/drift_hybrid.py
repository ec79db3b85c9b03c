"""Where does the fp32 loss of positivity enter?  The N = 200 all-measured stream runs three ways on the box:
  hip/hip   HIP predict + HIP update (control)
  hip/cpu   HIP predict, then the update in numpy fp32 written the way the HIP path factors it
            (W = Sigma H^T, Cholesky of S, V = W L^-T, Sigma -= V V^T, lower triangle mirrored)
  cpu/hip   numpy fp32 predict (the structured oracle's), then the HIP update
The min eigenvalue of Sigma is printed every `every` frames.  Test infrastructure: uses oracle/ as the CPU side.
usage: python tools/drift_hybrid.py [frames=2500] [every=250] [N=200]"""
import os, sys, time
import numpy as np
import scipy.linalg as sl
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import ekf_oracle as o
import oracle_worker
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import synthetic
import bench

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
every = int(sys.argv[2]) if len(sys.argv) > 2 else 250
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
T = np.float32
cfg = pkg.kinect_config()
px0, zs = synthetic.measurement_stream(cfg, N, frames, sigma_px=bench.SIGMA_Z_PX)
idx = list(range(N))


class Factored(o.StructuredFilter):
    def _update_block(self, indices, plane, z, h):
        Sg = self.Sigma
        W = self.sigma_Ht(indices, plane)
        St = self.H_times(W, indices, plane)
        p = St.shape[0]
        St[np.arange(p), np.arange(p)] += T(self.sigma_pixel_2)
        St = np.tril(St) + np.tril(St, -1).T
        L = np.linalg.cholesky(St).astype(T)
        V = sl.solve_triangular(L, W.T, lower=True).T.astype(T)
        y = sl.solve_triangular(L, (z - h), lower=True).astype(T)
        self.mu = self.mu + V @ y
        Sn = Sg - V @ V.T
        self.Sigma = np.tril(Sn) + np.tril(Sn, -1).T
        self.St, self.Kt = St, None


def make():
    g = pkg.VSlamFilter(cfg, capacity_features=N, dtype=T)
    g.setDt(1 / 30.0)
    f = Factored(o.Config.kinect(), T)
    f.dT = 1 / 30.0
    for (u, v) in px0:
        assert g.addFeature((u, v)) == 1 and f.add_feature(u, v) == 1
    return g, f


def to_cpu(g, f):
    f.Sigma = g.getFullSigma().astype(T)
    f.mu = g.getFullState().astype(T)


def to_hip(f, g):
    g.setSigmaBlock(f.Sigma, 0, 0)
    g.setFullState(f.mu)


def run(mode):
    g, f = make()
    t0 = time.time()
    for k in range(frames):
        if mode == "cpu/hip":
            to_cpu(g, f)
            oracle_worker.predict_no_St(f)
            to_hip(f, g)
            g.measure()
        else:
            g.predict()
        if k % every == 0:
            P = g.getFullSigma().astype(np.float64)
            w = np.linalg.eigvalsh(0.5 * (P + P.T))
            print(f"{mode} {k:6d} min eig {w[0]: .3e} 2nd {w[1]: .3e} max {w[-1]:.3e} [{time.time() - t0:.0f} s]", flush=True)
        z = zs[k].reshape(-1).astype(T)
        if mode == "hip/cpu":
            to_cpu(g, f)
            f.measure()
            f.update(z, idx)
            to_hip(f, g)
        else:
            try:
                g.update(z, idx)
                g.synchronize()
            except Exception as e:
                print(mode, "stopped at", k, e, flush=True)
                break


for mode in (sys.argv[4].split(",") if len(sys.argv) > 4 else ("hip/hip", "hip/cpu", "cpu/hip")):
    run(mode)
