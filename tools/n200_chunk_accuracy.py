"""N = 200 (configs[1]): the one-chunk update (explicit inverse of the WHOLE 4-block factor in the solve) against chunked
plans (EKF_OPT_PIPELINE = k: k equal chunks, sequential form) -- distance from the fp64 oracle on the fused_launches[200]
scenario (3 frames), and the step time on resident inputs.    python tools/n200_chunk_accuracy.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import ekf_oracle as o
from helpers import make_pair, gpu_state, relf, oracle_cfg
from __graft_entry__ import load_package
pkg = load_package()
import torch
from ekf_monoslam_amd import synthetic

N = 200
for pipe in (-1, 2, 4):
    ref, g = make_pair(N, np.float32)
    g.set_option(3, pipe)
    ref64 = o.build_scenario(o.StructuredFilter, oracle_cfg(), N, np.float64)
    ref64.mu = ref.mu.astype(np.float64).copy()
    ref64.Sigma = ref.Sigma.astype(np.float64).copy()
    out = []
    for k in range(3):
        ref.predict(); ref64.predict(); g.predict()
        vis = ref.visible_indices()
        z = o.synthetic_measurements(ref, vis, seed=500 + k)
        ref.update(z, vis); ref64.update(z.astype(np.float64), vis); g.update(z, vis)
        mu, S = gpu_state(g)
        out.append((relf(S, ref64.Sigma), relf(ref.Sigma, ref64.Sigma), relf(mu, ref64.mu)))
    plan = g.chunkPlan()
    g.close()
    # timing on resident inputs
    cfg = pkg.kinect_config()
    frames = 420
    px0, z = synthetic.measurement_stream(cfg, N, frames, sigma_px=0.5)
    d_z = torch.from_numpy(z.reshape(frames, -1)).cuda().contiguous()
    d_idx = torch.arange(N, dtype=torch.int32, device="cuda")
    f = pkg.VSlamFilter(cfg, capacity_features=N)
    f.setDt(1 / 30.0)
    f.set_option(3, pipe)
    for (u, v) in px0:
        f.addFeature((u, v))
    for k in range(10):
        f.predict(); f.update_device(d_z.data_ptr() + k * 8 * N, d_idx.data_ptr(), N)
    f.synchronize()
    t0 = time.perf_counter()
    for k in range(10, frames - 10):
        f.predict(); f.update_device(d_z.data_ptr() + k * 8 * N, d_idx.data_ptr(), N)
    f.synchronize()
    dt = (time.perf_counter() - t0) / (frames - 20)
    f.close()
    print(f"EKF_OPT_PIPELINE = {pipe:2d}  plan {plan}  {dt * 1e6:7.1f} us/step   |HIP - o64| Sigma per frame " +
          " ".join(f"{e[0]:.2e}" for e in out) + "   |o32 - o64| " + " ".join(f"{e[1]:.2e}" for e in out) +
          "   mu " + " ".join(f"{e[2]:.2e}" for e in out), flush=True)
