#!/usr/bin/env python3
"""fp32 positivity of a map whose features are ALL measured in EVERY frame.

The HIP fp32 filter, the fp32 structured oracle (the reference's formulation: explicit inverse, (I - K H) Sigma,
vR.cpp:1276-1279) and the fp64 oracle run the SAME stream; every `every` frames the smallest eigenvalue of the
innovation covariance S = H Sigma H^T + R (in exact arithmetic >= sigma_pixel^2 = 4) and of Sigma itself is recorded
for each of them.  The HIP path stops where its Cholesky meets a non-positive pivot (EKF_ERR_NUMERIC); the reference's
LU inverse (Eigen .inverse(), vR.cpp:1276) would not stop -- the oracle keeps going on an indefinite S.  A fourth
column runs the HIP filter with EKF_OPT_FEATURE_NOISE = 10000 (1e-8 added to every feature variance per predict).

usage: python tools/positivity_trace.py [N=200] [frames=6000] [every=100]
output: gpurun_out/r2_positivity_trace_N<N>.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import ekf_oracle as o  # noqa: E402
import oracle_worker  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def min_eigs(S_innov, Sigma):
    s = np.asarray(S_innov, np.float64)
    P = np.asarray(Sigma, np.float64)
    return float(np.linalg.eigvalsh(0.5 * (s + s.T)).min()), float(np.linalg.eigvalsh(0.5 * (P + P.T)).min())


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    pkg = load_package()
    from ekf_monoslam_amd import synthetic
    import bench
    cfg = pkg.kinect_config()
    px0, zs = synthetic.measurement_stream(cfg, N, frames, sigma_px=bench.SIGMA_Z_PX)
    idx = list(range(N))
    out_path = os.path.join(ROOT, "gpurun_out", f"r2_positivity_trace_N{N}.txt")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    fh = open(out_path, "w")

    def emit(line):
        print(line, flush=True)
        fh.write(line + "\n")
        fh.flush()

    hips = {}
    for name, floor in (("hip32", 0), ("hip32+noise", 1)):
        g = pkg.VSlamFilter(cfg, capacity_features=N, dtype=np.float32)
        g.setDt(1.0 / 30.0)
        for (u, v) in px0:
            assert g.addFeature((u, v)) == 1
        if floor:
            g.set_option(5, 10000)                              # EKF_OPT_FEATURE_NOISE: 1e-8 per predict
        hips[name] = g
    refs = {}
    for T in (np.float32, np.float64):
        r = o.StructuredFilter(o.Config.kinect(), T)
        r.dT = 1.0 / 30.0
        for (u, v) in px0:
            assert r.add_feature(u, v) == 1
        refs[T] = r
    emit(f"# N = {N}, M = N every frame, sigma_z = {bench.SIGMA_Z_PX} px, R = 4 I; min eigenvalue of S (exact >= 4) | of Sigma")
    emit("# frame   hip32: minS minSigma   hip32+noise: minS minSigma   oracle32: minS minSigma   oracle64: minS minSigma")
    dead = {}
    t0 = time.time()
    for k in range(frames):
        sample = (k % every == 0) or k == frames - 1
        row = [f"{k:6d}"]
        z = zs[k].reshape(-1)
        for name, g in hips.items():
            if name in dead:
                row.append("   (stopped)")
                continue
            g.predict()
            if sample:
                S = g.innovationCovariance(idx)
                a, b = min_eigs(S, g.getFullSigma())
                row.append(f"  {a: .3e} {b: .3e}")
            try:
                g.update(z, idx)
                g.synchronize()
                if not np.all(np.isfinite(g.getState())):
                    raise RuntimeError("non-finite state")
            except Exception as e:                                # EKF_ERR_NUMERIC: the Cholesky met a pivot <= 0
                dead[name] = k
                emit(f"# {name} stopped at frame {k}: {e}")
        for T, r in refs.items():
            oracle_worker.predict_no_St(r)
            if sample:
                a, b = min_eigs(r.innovation_covariance(idx), r.Sigma)
                row.append(f"  {a: .3e} {b: .3e}")
            r.update(z.astype(T), idx)
        if sample:
            emit(" ".join(row) + f"   [{time.time() - t0:.0f} s]")
        if len(dead) == 1 and "hip32" in dead and k > dead["hip32"] + 3 * every:
            break                                                 # three more samples past the stop of the plain path
    emit(f"# stops: {dead}")
    fh.close()


if __name__ == "__main__":
    main()
