"""fused_launches[200], frame 1: |HIP - o64| = 3e-5 of Sigma against 8e-6 for the fp32 oracle.  Decomposition on the GPU:
  (a) single-step error: the fp64 oracle's state after frame 0 (rounded to fp32) injected into the HIP filter, frame 1 run on it,
      against the fp64 oracle's frame 1 from the SAME injected state;
  (b) propagation: the HIP filter's OWN state after frame 0 handed to a fp64 oracle, whose exact frame 1 is compared with
      the HIP filter's frame 1 (its error on its own input) and with the fp64 trajectory (what the frame-0 difference grew into).
    python tools/n200_error_source.py"""
import os, sys, copy
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import ekf_oracle as o
from helpers import make_pair, gpu_state, relf, oracle_cfg
N = 200
ref, g = make_pair(N, np.float32)
ref64 = o.build_scenario(o.StructuredFilter, oracle_cfg(), N, np.float64)
ref64.mu = ref.mu.astype(np.float64).copy(); ref64.Sigma = ref.Sigma.astype(np.float64).copy()
# frame 0 on all three
ref.predict(); ref64.predict(); g.predict()
vis = ref.visible_indices(); z0 = o.synthetic_measurements(ref, vis, seed=500)
ref.update(z0, vis); ref64.update(z0.astype(np.float64), vis); g.update(z0, vis)
mu0, S0 = gpu_state(g)
print(f"frame 0: |HIP - o64| {relf(S0, ref64.Sigma):.2e}  |o32 - o64| {relf(ref.Sigma, ref64.Sigma):.2e}  asymmetry of o32 {np.abs(ref.Sigma - ref.Sigma.T).max() / np.abs(ref.Sigma).max():.1e}")
# (b) an fp64 oracle continuing from the HIP filter's own frame-0 state
own = copy.deepcopy(ref64)
own.mu = mu0.astype(np.float64).copy(); own.Sigma = S0.astype(np.float64).copy()
# (a) a second HIP filter and a second fp64 oracle, both from the fp64 state after frame 0 rounded to fp32
_, g2 = make_pair(N, np.float32)
inj = copy.deepcopy(ref64)
inj.mu = ref64.mu.astype(np.float32).astype(np.float64); inj.Sigma = ref64.Sigma.astype(np.float32).astype(np.float64)
g2.setFullState(inj.mu.astype(np.float32)); g2.setSigmaBlock(inj.Sigma.astype(np.float32))
# frame 1
ref.predict(); ref64.predict(); g.predict(); own.predict(); inj.predict(); g2.predict()
vis = ref.visible_indices(); z1 = o.synthetic_measurements(ref, vis, seed=501)
for r in (ref64, own, inj):
    r.update(z1.astype(np.float64), vis)
ref.update(z1, vis); g.update(z1, vis); g2.update(z1, vis)
mu1, S1 = gpu_state(g)
mu2, S2 = gpu_state(g2)
print(f"frame 1: |HIP - o64| {relf(S1, ref64.Sigma):.2e}   |o32 - o64| {relf(ref.Sigma, ref64.Sigma):.2e}")
print(f"  (a) single step from the injected fp64 state:            |HIP - exact| {relf(S2, inj.Sigma):.2e}")
print(f"  (b) HIP against the exact frame 1 of ITS OWN frame-0 state: {relf(S1, own.Sigma):.2e};  that exact result against the fp64 trajectory: {relf(own.Sigma, ref64.Sigma):.2e}")
d0 = S0.astype(np.float64) - ref64.Sigma if False else None
