import sys, numpy as np, scipy.linalg as sl
sys.path[:0] = ['/root/repo/oracle', '/root/repo/tests', '/root/repo']
import ekf_oracle as o
N = 200
cfg = o.Config.kinect()
ref = o.build_scenario(o.StructuredFilter, cfg, N, np.float32)
ref64 = o.build_scenario(o.StructuredFilter, cfg, N, np.float64)
ref64.mu = ref.mu.astype(np.float64).copy(); ref64.Sigma = ref.Sigma.astype(np.float64).copy()
def rel(a, b): return float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
T = np.float32
for k in range(2):
    ref.predict(); ref64.predict()
    vis = ref.visible_indices()
    z = o.synthetic_measurements(ref, vis, seed=500 + k)
    Sig = ref.Sigma.copy()
    Sig64 = Sig.astype(np.float64)
    W = ref.sigma_Ht(vis, False); St = ref.H_times(W, vis, False)
    St[np.arange(St.shape[0]), np.arange(St.shape[0])] += T(ref.sigma_pixel_2)
    # exact (fp64) update of THIS fp32 input: the reference point for formulation errors
    r2 = o.StructuredFilter(cfg, np.float64); 
    W64 = W.astype(np.float64); 
    # fp64 W and S from the fp32 Sigma
    ref_tmp = ref64
    L = np.linalg.cholesky(St).astype(T)
    V = sl.solve_triangular(L, W.T, lower=True).T.astype(T)
    St64 = St.astype(np.float64)
    L64 = np.linalg.cholesky(St64); V64 = sl.solve_triangular(L64, W64.T, lower=True).T
    exact = Sig64 - V64 @ V64.T                       # exact downdate for the fp32-rounded W, S
    f = slice(14, None)
    A = (Sig - (V @ V.T).astype(T)).astype(T)                       # all fp32
    B = (Sig - (V64.astype(T) @ V64.astype(T).T).astype(T)).astype(T)   # V exact (rounded), product fp32
    C = (Sig64 - V.astype(np.float64) @ V.astype(np.float64).T)     # V fp32, product exact
    Kt = (W @ np.linalg.inv(St)).astype(T); HS = ref.H_times(Sig, vis, False)
    D = (Sig - (Kt @ HS).astype(T)).astype(T)
    print(f"frame {k}: cond(S) {np.linalg.cond(St64):.2e}  vs exact-of-same-input (features): all-fp32 VV^T {rel(A[f,f], exact[f,f]):.2e}  exact V, fp32 product {rel(B[f,f], exact[f,f]):.2e}  "
          f"fp32 V, exact product {rel(C[f,f], exact[f,f]):.2e}  K(HS) fp32 {rel(D[f,f], exact[f,f]):.2e}   |V-V64|/|V64| {rel(V, V64):.2e}")
    ref.update(z, vis); ref64.update(z.astype(np.float64), vis)

# sensitivity: the exact (fp64) frame-1 update applied to the fp64 state after frame 0 plus a symmetric perturbation of the
# size of the frame-0 error (1.8e-6 relative Frobenius)
ref64 = o.build_scenario(o.StructuredFilter, cfg, N, np.float64)
base = o.build_scenario(o.StructuredFilter, cfg, N, np.float32)
ref64.mu = base.mu.astype(np.float64).copy(); ref64.Sigma = base.Sigma.astype(np.float64).copy()
ref64.predict(); base.predict(); vis = ref64.visible_indices(); z0 = o.synthetic_measurements(base, vis, seed=500)
ref64.update(z0.astype(np.float64), vis)
import copy
rng = np.random.default_rng(0)
for scale in (1.8e-6, 1.8e-7):
    p = copy.deepcopy(ref64)
    E = rng.standard_normal(p.Sigma.shape); E = (E + E.T) / 2
    E *= np.abs(p.Sigma)                                  # element-wise relative perturbation
    E *= scale * np.linalg.norm(p.Sigma) / np.linalg.norm(E)
    p.Sigma = p.Sigma + E
    q = copy.deepcopy(ref64)
    for r in (p, q):
        r.predict()
    vis = q.visible_indices(); z1 = o.synthetic_measurements(q, vis, seed=501)
    for r in (p, q):
        r.update(z1.astype(np.float64), vis)
    print(f"symmetric input perturbation {scale:.1e} -> output deviation after the exact frame-1 update {rel(p.Sigma, q.Sigma):.2e} (features {rel(p.Sigma[14:,14:], q.Sigma[14:,14:]):.2e})")
