"""Shared helpers of the parity tests: build an oracle filter and a HIP filter on identical
inputs (SURVEY.md 8d scenario: kinect intrinsics, seeds 1234/1235, features inserted through
the add-feature math)."""
import numpy as np

import ekf_oracle as o
from __graft_entry__ import load_package


def relf(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) /
                 max(np.linalg.norm(np.asarray(b, np.float64)), 1e-300))


def gpu_filter(n_feat, dtype, capacity=None, cfg_name="kinect", camera_dim=14, mfma=True):
    pkg = load_package()
    cfg = pkg.kinect_config() if cfg_name == "kinect" else pkg.sim_config()
    f = pkg.VSlamFilter(cfg, capacity_features=capacity or max(n_feat, 1), dtype=dtype, camera_dim=camera_dim)
    f.set_option(1, 1 if mfma else 0)     # EKF_OPT_USE_MFMA
    return f


def oracle_cfg(cfg_name="kinect"):
    return o.Config.kinect() if cfg_name == "kinect" else o.Config.sim()


def make_pair(n_feat, dtype, flavour=None, cfg_name="kinect", camera_dim=14, mfma=True, capacity=None,
              inject=True):
    """Oracle scenario + HIP filter with the same features.  With inject=True the oracle's mu and
    Sigma are copied into the HIP filter so later steps are compared on bit-identical inputs."""
    flavour = flavour or o.StructuredFilter
    cfg = oracle_cfg(cfg_name)
    ref = o.build_scenario(flavour, cfg, n_feat, dtype, camera_dim=camera_dim)
    g = gpu_filter(n_feat, dtype, capacity, cfg_name, camera_dim, mfma)
    g.setDt(ref.dT)
    full = g.getFullState()
    full[7:13] = ref.mu[7:13]
    g.setFullState(full)
    for (u, v) in o.synthetic_pixels(cfg, n_feat):
        assert g.addFeature((u, v)) == 1
    if inject:
        g.setFullState(ref.mu)
        g.setSigmaBlock(ref.Sigma)
    return ref, g


def gpu_state(g):
    return g.getFullState(), g.getFullSigma()
