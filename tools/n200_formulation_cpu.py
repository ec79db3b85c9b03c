"""CPU only (numpy, no GPU): whose error is the fused_launches[200] frame-1 site?  The symmetric downdate Sigma - V V^T with
V = W L^-T (what the HIP path computes) in plain fp32 numpy, with and without the explicit inverse of L, against the
reference's Sigma - K (H Sigma) in fp32 (the fp32 oracle), all measured against the fp64 oracle on the same stream.
    python tools/n200_formulation_cpu.py"""
import sys, numpy as np, scipy.linalg as sl
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests'), ROOT]
import ekf_oracle as o
N = 200
cfg = o.Config.kinect()
ref = o.build_scenario(o.StructuredFilter, cfg, N, np.float32)
ref64 = o.build_scenario(o.StructuredFilter, cfg, N, np.float64)
ref64.mu = ref.mu.astype(np.float64).copy(); ref64.Sigma = ref.Sigma.astype(np.float64).copy()
def rel(a, b): return float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
for k in range(3):
    ref.predict(); ref64.predict()
    vis = ref.visible_indices()
    z = o.synthetic_measurements(ref, vis, seed=500 + k)
    # the Cholesky / V V^T formulation in fp32 on the fp32 oracle's state (what the HIP path computes, plain numpy order)
    T = np.float32
    Sig = ref.Sigma.copy()
    W = ref.sigma_Ht(vis, False)
    St = ref.H_times(W, vis, False)
    R = np.eye(St.shape[0], dtype=T) * T(ref.sigma_pixel_2)
    St = (St + R).astype(T)
    L = np.linalg.cholesky(St.astype(T)).astype(T)
    V = sl.solve_triangular(L, W.T, lower=True).T.astype(T)
    S_chol = (Sig - (V @ V.T).astype(T)).astype(T)
    # the same with the explicit inverse of L
    Z = sl.solve_triangular(L, np.eye(L.shape[0], dtype=T), lower=True).T.astype(T)   # L^-T
    V2 = (W @ Z).astype(T)
    S_inv = (Sig - (V2 @ V2.T).astype(T)).astype(T)
    ref.update(z, vis); ref64.update(z.astype(np.float64), vis)
    # quaternion normalisation differs between S_chol and ref.Sigma (applied inside update) -> compare the feature block only
    f = slice(14, None)
    # fp64 truth of the un-normalised update is not kept; use the feature block (the normalisation only touches rows/cols 3..6)
    print(f"frame {k}: |o32-o64| {rel(ref.Sigma, ref64.Sigma):.2e}  features only: o32 {rel(ref.Sigma[f, f], ref64.Sigma[f, f]):.2e}  "
          f"chol fp32 {rel(S_chol[f, f], ref64.Sigma[f, f]):.2e}  chol+explicit inverse {rel(S_inv[f, f], ref64.Sigma[f, f]):.2e}")
