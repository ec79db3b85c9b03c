"""Known-answer tests that pin the oracle to facts readable off the reference source
(SURVEY.md 8c item 1).  Each case cites the reference lines it is derived from."""
import numpy as np
import pytest

import ekf_oracle as o

T = np.float64


def test_vec2quat_zero_is_identity():            # vR.cpp:1397-1399
    assert np.array_equal(o.vec2quat([0, 0, 0], T), [1, 0, 0, 0])


def test_vec2quat_axis_angle():                  # vR.cpp:1393-1394
    q = o.vec2quat([0, 0, np.pi], T)
    assert np.allclose(q, [0, 0, 0, 1], atol=1e-15)


def test_quat2rot_identity():                    # vR.cpp:1416-1418
    assert np.array_equal(o.quat2rot([1, 0, 0, 0], T), np.eye(3))


def test_quat2rot_is_rotation_and_matches_product():
    rng = np.random.default_rng(0)
    q1 = rng.normal(size=4); q1 /= np.linalg.norm(q1)
    q2 = rng.normal(size=4); q2 /= np.linalg.norm(q2)
    R1, R2 = o.quat2rot(q1, T), o.quat2rot(q2, T)
    assert np.allclose(R1 @ R1.T, np.eye(3), atol=1e-14)
    assert np.isclose(np.linalg.det(R1), 1.0)
    # Hamilton product composes rotations (vR.cpp:1423-1460)
    assert np.allclose(o.quat2rot(o.quat_product(q1, q2, T), T), R1 @ R2, atol=1e-14)
    # complement = inverse rotation (vR.cpp:1568-1572)
    assert np.allclose(o.quat2rot(o.quat_complement(q1, T), T), R1.T, atol=1e-14)


def test_system_model_jacobian_at_zero_rate():   # vR.cpp:1497-1504, 1521, 1531-1532
    x = np.zeros(13); x[3:7] = [0.5, 0.5, -0.5, 0.5]
    dT = 0.25
    Ft = o.system_model_jacobian(x, dT, [0, 0, 0], T)
    assert np.array_equal(Ft[3:7, 3:7], np.eye(4))
    assert np.allclose(Ft[3:7, 10:13], 0.5 * dT * o.upsilon(x[3:7], T)[:, 1:4])
    assert np.array_equal(Ft[0:3, 7:10], dT * np.eye(3))
    mask = np.ones((13, 13), bool)
    mask[3:7, 3:7] = mask[3:7, 10:13] = mask[0:3, 7:10] = False
    assert np.array_equal(Ft[mask], np.eye(13)[mask])


def test_project_principal_axis():               # cam.cpp:78-108
    cam = o.CamModel(o.Config.kinect(), T)
    hd, J = cam.project([0, 0, 1])
    assert np.allclose(hd, [cam.u0, cam.v0])
    assert np.allclose(J, [[cam.fx, 0, 0], [0, cam.fy, 0]])


def test_unproject_zero_distortion_is_exact():   # cam.cpp:165-174
    cfg = o.Config.sim()
    cam = o.CamModel(cfg, T)
    hC, J = cam.unproject([100.0, 50.0])
    assert np.allclose(hC, [(100.0 - cam.u0) / cam.fx, (50.0 - cam.v0) / cam.fy, 1.0])
    assert np.allclose(J, [[1 / cam.fx, 0], [0, 1 / cam.fy], [0, 0]])


def test_unproject_inverts_project_with_distortion():
    cam = o.CamModel(o.Config.kinect(), T)
    for px in ([20.0, 30.0], [300.0, 200.0], [160.0, 120.0]):
        hC, _ = cam.unproject(px)
        hd, _ = cam.project(hC)
        assert np.allclose(hd, px, atol=1e-9)


def test_initial_state_constants():              # vR.cpp:146, 163-164, 180, 211-216
    f = o.DenseFilter(o.Config(), np.float32)
    assert f.mu.shape == (14,)
    assert np.allclose(f.mu, [0, 0, 0, 0, 0, -0.707106781, 0.707106781, 0, 0, 0, 0, 0, 0, 1])
    d = np.diag(f.Sigma)
    assert np.allclose(d[:7], 4e-10) and np.isclose(d[13], 0.09)
    assert np.allclose(d[7:13], 1.6e-7)
    assert np.count_nonzero(f.Sigma - np.diag(d)) == 0


def test_process_noise_doubles_without_control():  # vR.cpp:202, 463-473
    f = o.DenseFilter(o.Config.kinect(), T)
    f.dT = 0.1
    _, Qc = f._motion([0, 0, 0], [0, 0, 0], True)
    _, Qn = f._motion([0, 0, 0], [0, 0, 0], False)
    assert np.allclose(Qn, 2 * Qc)
    assert np.isclose(Qc[7, 7], (0.03 ** 2) / 0.01)


def test_normalize_unit_quaternion_projects_block():  # vR.cpp:1634-1636
    f = o.DenseFilter(o.Config(), T)
    rng = np.random.default_rng(1)
    A = rng.normal(size=(14, 14)); f.Sigma = A @ A.T
    q = f.mu[3:7].copy()
    S0 = f.Sigma.copy()
    f.normalize_quaternion()
    P = np.eye(4) - np.outer(q, q)
    assert np.allclose(f.Sigma[3:7, 3:7], P @ S0[3:7, 3:7] @ P.T, atol=1e-12)
    assert np.allclose(f.mu[3:7], q)


def test_add_feature_rejects_border_pixels():    # vR.cpp:314, 1644-1652
    cfg = o.Config.kinect()
    f = o.DenseFilter(cfg, T)
    assert f.add_feature(3.0, 100.0) == 0
    assert f.add_feature(cfg.window_size // 2, 100.0) == 0          # strict '>'
    assert f.add_feature(100.0, cfg.image_height - cfg.window_size // 2) == 0
    assert f.num_features() == 0 and f.n == 14
    assert f.add_feature(100.0, 100.0) == 1
    assert f.n == 20 and f.features[0].position_in_state == 14


def test_add_feature_uses_sigma_rho_unsquared():  # vR.cpp:365
    cfg = o.Config.kinect()
    f = o.DenseFilter(cfg, T)
    f.add_feature(100.0, 100.0)
    assert np.isclose(f.Sigma[19, 19], cfg.sigma_rho_0)
    assert np.isclose(f.mu[19], cfg.rho_0)
    assert np.allclose(f.mu[14:17], f.mu[0:3])


def test_round_trip_add_then_measure_returns_pixel():  # vR.cpp:326-346 then 525-528
    cfg = o.Config.kinect()
    f = o.DenseFilter(cfg, T)
    px = o.synthetic_pixels(cfg, 12, seed=7)
    for u, v in px:
        f.add_feature(u, v)
    f.measure()
    for ft, p in zip(f.features, px):
        assert ft.is_in_innovation
        assert np.allclose(ft.h, p, atol=1e-8)


def test_negative_rho_flags_removal():           # vR.cpp:517-522
    f = o.DenseFilter(o.Config.kinect(), T)
    f.add_feature(100.0, 100.0)
    f.mu[19] = -0.1
    f.measure()
    assert f.features[0].remove_flag and not f.features[0].is_in_innovation


def test_remove_feature_shifts_positions():      # vR.cpp:408-419
    cfg = o.Config.kinect()
    f = o.build_scenario(o.DenseFilter, cfg, 4, T)
    S0, mu0 = f.Sigma.copy(), f.mu.copy()
    f.remove_feature(1)
    keep = np.r_[0:20, 26:38]
    assert np.array_equal(f.mu, mu0[keep])
    assert np.array_equal(f.Sigma, S0[np.ix_(keep, keep)])
    assert [ft.position_in_state for ft in f.features] == [14, 20, 26]


def test_covariance_parameter():                 # vR.cpp:854-855
    f = o.build_scenario(o.DenseFilter, o.Config.kinect(), 3, T)
    f.predict()
    assert np.isclose(f.covariance_parameter(), np.trace(f.Sigma[:7, :7]))


def test_camera_dim_13_variant_has_no_scale_element():
    f = o.build_scenario(o.StructuredFilter, o.Config.kinect(), 5, T, camera_dim=13)
    assert f.n == 13 + 30
    f.predict()
    z = o.synthetic_measurements(f, f.visible_indices())
    f.update(z)
    assert np.all(np.isfinite(f.Sigma))


def test_glibc_rand_matches_the_c_library():
    """The RANSAC draws of the reference are srand(time(NULL)) / rand() (vR.cpp:970, 989): the oracle's (and the
    library's) replay generator must BE glibc's rand() -- checked against the C library of this machine."""
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    for seed in (1, 7, 42, 123456789, 2 ** 31 + 5, 2 ** 32 - 1):
        libc.srand(ctypes.c_uint(seed))
        want = [libc.rand() for _ in range(400)]
        g = o.GlibcRand(seed)
        assert [g.rand() for _ in range(400)] == want


def test_points_table_layout_known_answer():
    """getPointsFeatures (RosVSLAMRansac.cpp:340-418) from the source alone: real_index counts from 1 (vR.cpp:148,
    318-319), the table has (real_index of the last live patch) + 1 rows (:350-352), inverse-depth rows are zero
    (:363-375), an XYZ row is [mu * map_scale | the 3x3 block row by row] (:376-388), and a removed XYZ patch with
    n_find > 5 keeps the values of the moment of its removal (vR.cpp:394-404, :406-414)."""
    f = o.build_scenario(o.StructuredFilter, o.Config.kinect(), 6, np.float64)
    assert [ft.real_index for ft in f.features] == [1, 2, 3, 4, 5, 6] and f.patchnumbre == 7
    for i in (1, 3):
        p = f.features[i].position_in_state
        f.Sigma[p + 5, p + 5] = 1e-9
    assert f.convert2xyz_if_linear_all() == 2
    f.mu[13] = 2.0                                        # map_scale
    t = o.get_points_features(f)
    assert t.shape == (7, 12) and not t[0].any() and not t[1].any() and not t[3].any()
    p = f.features[1].position_in_state
    assert np.array_equal(t[2, :3], f.mu[p:p+3] * 2.0) and np.array_equal(t[2, 3:], f.Sigma[p:p+3, p:p+3].reshape(-1))
    f.features[1].n_find = 6
    xyz, cov = f.mu[p:p+3].copy(), f.Sigma[p:p+3, p:p+3].copy()
    f.remove_feature(1)
    f.remove_feature(2)                                   # the other XYZ patch (real_index 4), n_find = 1: not kept
    assert [d[0] for d in f.deleted_patches] == [2]
    f.mu[0:3] += 1.0                                      # whatever happens to the filter afterwards
    t = o.get_points_features(f)
    assert t.shape == (7, 12) and np.array_equal(t[2, :3], xyz * 2.0) and np.array_equal(t[2, 3:], cov.reshape(-1))
    assert not t[4].any()
    f.remove_feature(len(f.features) - 1)                 # last live patch gone: the table ends at real_index 5
    assert o.get_points_features(f).shape == (6, 12)
    assert f.add_feature(100.0, 100.0) == 1 and f.features[-1].real_index == 7
