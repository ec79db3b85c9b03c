"""Faithful-dense vs structured oracle agreement, the EKF
identities the update must satisfy, and the resize paths."""
import numpy as np
import pytest

import ekf_oracle as o


def relf(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def run(flavour, dtype, n_feat, frames=2, cfg=None, plane=False):
    cfg = cfg or o.Config.kinect()
    f = o.build_scenario(flavour, cfg, n_feat, dtype)
    for k in range(frames):
        f.predict()
        vis = f.visible_indices()
        z = o.synthetic_measurements(f, vis, seed=1235 + k)
        f.update(z, vis, plane=plane)
    return f


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 2e-5)])
@pytest.mark.parametrize("n_feat", [20, 60])
def test_dense_matches_structured(dtype, tol, n_feat):
    a = run(o.DenseFilter, dtype, n_feat)
    b = run(o.StructuredFilter, dtype, n_feat)
    assert relf(a.Sigma, b.Sigma) < tol
    assert relf(a.mu, b.mu) < tol
    assert relf(a.St, b.St) < tol
    assert relf(a.Kt, b.Kt) < tol * 50


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_batched_add_is_the_sequential_add(dtype):
    """`StructuredFilter.add_features` (one preallocated covariance; what the N = 1000 parity tests build their oracle
    with) against N sequential `add_feature` calls, and against the reference-shaped dense add (Js S' Js^T): bit-identical
    to the former, to rounding equal to the latter.  A pixel outside the image is skipped by both."""
    cfg = o.Config.kinect()
    a = o.build_scenario(o.StructuredFilter, cfg, 50, dtype)
    b = o.build_scenario(o.StructuredFilter, cfg, 50, dtype, batched_add=True)
    assert np.array_equal(a.mu, b.mu) and np.array_equal(a.Sigma, b.Sigma)
    assert [f.position_in_state for f in a.features] == [f.position_in_state for f in b.features]
    d = o.build_scenario(o.DenseFilter, cfg, 50, dtype)
    assert relf(b.Sigma, d.Sigma) < (1e-12 if dtype == np.float64 else 2e-6)
    px = list(o.synthetic_pixels(cfg, 5)) + [(1.0, 1.0)] + list(o.synthetic_pixels(cfg, 3, seed=9))
    c1, c2 = o.StructuredFilter(cfg, dtype), o.StructuredFilter(cfg, dtype)
    n1 = sum(c1.add_feature(u, v) for (u, v) in px)
    assert c2.add_features(px) == n1 == 8
    assert np.array_equal(c1.Sigma, c2.Sigma) and np.array_equal(c1.mu, c2.mu)
    # further adds on top of a batched map (what the resize tests do)
    assert a.add_feature(100.0, 100.0) == 1 and b.add_feature(100.0, 100.0) == 1
    assert np.array_equal(a.Sigma, b.Sigma)


def test_dense_matches_structured_n200_single_frame():
    a = run(o.DenseFilter, np.float64, 200, frames=1)
    b = run(o.StructuredFilter, np.float64, 200, frames=1)
    assert a.n == 14 + 6 * 200
    assert relf(a.Sigma, b.Sigma) < 1e-11
    assert relf(a.mu, b.mu) < 1e-12


def test_plane_pseudo_measurement():              # vR.cpp:1245-1274
    a = run(o.DenseFilter, np.float64, 15, frames=1, plane=True)
    b = run(o.StructuredFilter, np.float64, 15, frames=1, plane=True)
    assert a.St.shape[0] == 2 * 15 + 3
    assert relf(a.Sigma, b.Sigma) < 1e-12
    # the pseudo-measurement pins y, qx, qz near zero with R = 1e-5
    free = run(o.DenseFilter, np.float64, 15, frames=1, plane=False)
    assert abs(a.mu[1]) <= abs(free.mu[1]) + 1e-9


def test_update_equals_information_form():
    """(I-KH) Sigma == (Sigma^-1 + H^T R^-1 H)^-1 for the exact gain (fp64)."""
    f = o.build_scenario(o.DenseFilter, o.Config.kinect(), 10, np.float64)
    f.predict()
    vis = f.visible_indices()
    H = f.dense_H(vis)
    P = f.Sigma.copy()
    z = o.synthetic_measurements(f, vis)
    q = f.mu[3:7].copy()
    f._update_block(vis, False, z, f.stacked_h(vis))
    # P has tiny eigenvalues (4e-10): compare through the Joseph identity instead
    K = f.Kt
    R = 4.0 * np.eye(H.shape[0])
    joseph = (np.eye(f.n) - K @ H) @ P @ (np.eye(f.n) - K @ H).T + K @ R @ K.T
    assert relf(f.Sigma, joseph) < 1e-9


def test_first_frame_sizes_and_symmetry_drift():
    f = run(o.DenseFilter, np.float32, 20, frames=1)
    assert f.St.shape == (40, 40) and f.Kt.shape == (134, 40)
    # the reference never symmetrises: asymmetry stays at rounding level
    assert relf(f.Sigma, f.Sigma.T) < 1e-5


@pytest.mark.parametrize("flavour", [o.DenseFilter, o.StructuredFilter])
def test_convert_to_xyz_paths(flavour):           # vR.cpp:741-780
    f = o.build_scenario(flavour, o.Config.kinect(), 8, np.float64)
    f.predict()
    z = o.synthetic_measurements(f, f.visible_indices())
    f.update(z)
    assert f.convert2xyz_if_linear_all() == 0     # fresh features are far from linear
    pos = f.features[2].position_in_state
    f.Sigma[pos + 5, pos + 5] = 1e-9
    n0 = f.n
    y_before, C_before = f.feature_xyz(2)
    assert f.convert2xyz_if_linear(2)
    assert f.n == n0 - 3 and f.features[2].coding == o.XYZ
    assert [ft.position_in_state for ft in f.features] == [14, 20, 26, 29, 35, 41, 47, 53]
    y_after, C_after = f.feature_xyz(2)
    assert np.allclose(y_before, y_after)
    assert np.allclose(C_before, C_after, rtol=1e-9, atol=1e-14)
    f.predict()                                   # XYZ branch of the measurement loop
    assert f.features[2].Hf.shape == (2, 3)
    z = o.synthetic_measurements(f, f.visible_indices())
    f.update(z)
    assert np.all(np.isfinite(f.Sigma))


def test_convert_dense_matches_structured():
    fs = []
    for fl in (o.DenseFilter, o.StructuredFilter):
        f = o.build_scenario(fl, o.Config.kinect(), 6, np.float64)
        f.predict()
        f.update(o.synthetic_measurements(f, f.visible_indices()))
        for i in (1, 4):
            pos = f.features[i].position_in_state
            f.Sigma[pos + 5, pos + 5] = 1e-9
        assert f.convert2xyz_if_linear_all() == 2
        f.remove_feature(0)
        f.predict()
        f.update(o.synthetic_measurements(f, f.visible_indices()))
        fs.append(f)
    assert relf(fs[0].Sigma, fs[1].Sigma) < 1e-12
    assert relf(fs[0].mu, fs[1].mu) < 1e-12
