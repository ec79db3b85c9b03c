#!/usr/bin/env python3
"""Generates the golden vectors of BASELINE config 1 (N = 20 inverse-depth features, n = 134,
one predict + update) from the CPU oracle, in fp64 and fp32.

The reference ships no fixtures and cannot be built here, so these vectors pin
the ORACLE'S outputs (regression guard + a GPU-box check that needs no oracle run), not the
reference's.  Inputs follow SURVEY.md 8d: conf_kinect.cfg with scale 2, pixels seed 1234,
measurement noise seed 1235, v = (0.3, 0, 0) m/s, w = (0, 0.05, 0) rad/s, dT = 1/30 s.

    python tests/golden/make_golden.py        # rewrites tests/golden/config1_n20_{f64,f32}.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ekf_oracle as o  # noqa: E402


def run(dtype):
    cfg = o.Config.kinect()
    pixels = o.synthetic_pixels(cfg, 20, seed=1234)
    f = o.build_scenario(o.DenseFilter, cfg, 20, dtype)
    out = {"pixels": pixels, "dT": np.float64(f.dT), "mu_added": f.mu.copy(), "Sigma_added": f.Sigma.copy()}
    f.predict()
    vis = f.visible_indices()
    out.update(mu_pred=f.mu.copy(), Sigma_pred=f.Sigma.copy(), visible=np.array(vis, np.int32),
               h=np.stack([ft.h for ft in f.features]), Hc=np.stack([ft.Hc for ft in f.features]),
               Hf=np.stack([ft.Hf for ft in f.features]), St=f.St.copy(), Ft=f.Ft.copy())
    z = o.synthetic_measurements(f, vis, seed=1235)
    f.update(z, vis)
    out.update(z=z, Kt=f.Kt.copy(), mu_upd=f.mu.copy(), Sigma_upd=f.Sigma.copy())
    return out


if __name__ == "__main__":
    for name, dt in (("f64", np.float64), ("f32", np.float32)):
        path = os.path.join(HERE, f"config1_n20_{name}.npz")
        np.savez_compressed(path, **run(dt))
        print("wrote", path, os.path.getsize(path), "bytes")
