#!/usr/bin/env python3
"""Which side carries the fp32 error of a long stream?  (VERDICT r1, weak #1 / next #1a)

N = 200, fp32, the bench's 1000-frame stream.  The HIP filter runs the whole stream; at a few points the fp32 AND the
fp64 structured oracle are both re-started from the HIP state (mu and the exactly symmetric Sigma injected) and run K
frames beside it.  Per probed frame: |HIP - oracle32|, |HIP - oracle64|, |oracle32 - oracle64| (rel-Frobenius on Sigma,
and on mu): the fp64 run from the same start is the truth of the segment, so the last two columns say whose rounding
the first column is made of.  Output: gpurun_out/r2_stream_parity_probe.txt (copied to profiles/).
usage: python tools/stream_parity_probe.py [N] [K]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ekf_oracle as o  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def relf(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) /
                 max(np.linalg.norm(np.asarray(b, np.float64)), 1e-300))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    frames = 1000
    starts = [0, 100, 300, 600, 900, 975]
    pkg = load_package()
    from ekf_monoslam_amd import synthetic
    cfg = pkg.kinect_config()
    px0, zs = synthetic.measurement_stream(cfg, N, frames, sigma_px=0.5)
    g = pkg.VSlamFilter(cfg, capacity_features=N, dtype=np.float32)
    g.setDt(1.0 / 30.0)
    for (u, v) in px0:
        assert g.addFeature((u, v)) == 1
    refs = {}
    for T in (np.float32, np.float64):
        r = o.StructuredFilter(o.Config.kinect(), T)
        r.dT = 1.0 / 30.0
        for (u, v) in px0:
            assert r.add_feature(u, v) == 1
        refs[T] = r
    idx = np.arange(N, dtype=np.int32)
    out = ["# frame  S:hip-o32  S:hip-o64  S:o32-o64   mu:hip-o32  mu:hip-o64  mu:o32-o64   |Sigma|_F  cond-proxy(max diag/min diag)"]
    live = False
    left = 0
    for k in range(frames):
        if k in starts:
            mu, S = g.getFullState(), g.getFullSigma()
            for T, r in refs.items():
                r.mu, r.Sigma = mu.astype(T), S.astype(T)
            live, left = True, K
            out.append(f"# --- oracles re-started from the HIP state at frame {k}")
        g.predict()
        z = zs[k].reshape(-1)
        if live:
            for T, r in refs.items():
                r.predict()
                assert r.visible_indices() == list(range(N))
                r.update(z.astype(T), list(range(N)))
        g.update(z, idx)
        if live:
            mu, S = g.getFullState(), g.getFullSigma()
            r32, r64 = refs[np.float32], refs[np.float64]
            d = np.diag(S.astype(np.float64))
            out.append("%5d  %.3e  %.3e  %.3e   %.3e  %.3e  %.3e   %.3e  %.2e" % (
                k, relf(S, r32.Sigma), relf(S, r64.Sigma), relf(r32.Sigma, r64.Sigma),
                relf(mu, r32.mu), relf(mu, r64.mu), relf(r32.mu, r64.mu), np.linalg.norm(S), d.max() / d.min()))
            left -= 1
            live = left > 0
    g.synchronize()
    text = "\n".join(out)
    print(text)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "r2_stream_parity_probe.txt"), "w").write(text + "\n")


if __name__ == "__main__":
    main()
