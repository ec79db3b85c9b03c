"""Error of the fp32 update against the fp64 oracle: exact fp32 MFMA path vs EKF_OPT_SPLIT_BF16 (N = 530, one frame)."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "oracle"), os.path.join(R, "tests")]
import ekf_oracle as o
from helpers import make_pair, gpu_state, relf, oracle_cfg
n_feat = int(sys.argv[1]) if len(sys.argv) > 1 else 530
ref, g0 = make_pair(n_feat, np.float32, capacity=n_feat)
ref64 = o.build_scenario(o.StructuredFilter, oracle_cfg(), n_feat, np.float64)
_, g1 = make_pair(n_feat, np.float32, capacity=n_feat)
g1.set_option(4, 1)
ref64.predict(); vis = ref64.visible_indices()
z = o.synthetic_measurements(ref64, vis, seed=1235, sigma=0.5)
ref64.update(z, vis)
ref.predict(); ref.update(z.astype(np.float32), vis)
print("fp32 numpy oracle  vs fp64: mu %.3e  Sigma %.3e" % (relf(ref.mu, ref64.mu), relf(ref.Sigma, ref64.Sigma)))
for name, g in (("exact fp32 MFMA  ", g0), ("split 3 x bf16   ", g1)):
    g.setFullState(o.build_scenario(o.StructuredFilter, oracle_cfg(), n_feat, np.float32).mu)
    g.setSigmaBlock(o.build_scenario(o.StructuredFilter, oracle_cfg(), n_feat, np.float32).Sigma)
    g.predict(); g.update(z.astype(np.float32), vis); g.synchronize()
    mu, S = gpu_state(g)
    print("%s vs fp64: mu %.3e  Sigma %.3e  max|S-S^T|/max|S| %.1e" % (name, relf(mu, ref64.mu), relf(S, ref64.Sigma), np.abs(S - S.T).max() / np.abs(S).max()))
