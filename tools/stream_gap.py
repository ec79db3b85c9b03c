"""Per-frame distances on the N=200 stream: HIP fp32 vs oracle fp32 vs oracle fp64 (diagnostic)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ekf_oracle as o
from helpers import make_pair, gpu_state, relf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ref, g = make_pair(N, np.float32)
ref64 = o.build_scenario(o.StructuredFilter, o.Config.kinect(), N, np.float64)
for k in range(40):
    ref.predict(); g.predict(); ref64.predict()
    vis = ref.visible_indices()
    z = o.synthetic_measurements(ref, vis, seed=2000 + k, sigma=0.5)
    ref.update(z, vis); g.update(z, vis); ref64.update(z.astype(np.float64), vis)
    mu, S = gpu_state(g)
    print(k, "hip-o32 mu %.2e S %.2e | o32-o64 mu %.2e S %.2e | hip-o64 mu %.2e S %.2e | vis %d" % (
        relf(mu, ref.mu), relf(S, ref.Sigma), relf(ref.mu, ref64.mu), relf(ref.Sigma, ref64.Sigma), relf(mu, ref64.mu), relf(S, ref64.Sigma), len(vis)))
