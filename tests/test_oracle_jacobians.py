"""Finite-difference checks (fp64) of every analytic Jacobian the oracle restates
(SURVEY.md 8c item 2): a2, a4, a5, a6, a12, a14."""
import numpy as np

import ekf_oracle as o

T = np.float64


def fd(fun, x, eps=1e-6):
    x = np.asarray(x, dtype=T)
    f0 = np.asarray(fun(x))
    J = np.empty((f0.size, x.size))
    for i in range(x.size):
        xp, xm = x.copy(), x.copy()
        xp[i] += eps
        xm[i] -= eps
        J[:, i] = (np.asarray(fun(xp)) - np.asarray(fun(xm))).ravel() / (2 * eps)
    return J


def test_projection_jacobian():                  # cam.cpp:68-111
    cam = o.CamModel(o.Config.kinect(), T)
    for hC in ([0.3, -0.2, 2.0], [-1.0, 0.7, 3.5], [0.01, 0.02, 0.9]):
        _, J = cam.project(hC)
        Jn = fd(lambda x: cam.project(x, False)[0], hC)
        assert np.allclose(J, Jn, rtol=1e-6, atol=1e-6)


def test_distortion_jacobian():                  # cam.cpp:18-47
    cam = o.CamModel(o.Config(), T)             # default wide-angle set, strong distortion

    def distort(hn):
        hd, _ = cam.project([hn[0], hn[1], 1.0], False)
        return [(hd[0] - cam.u0) / cam.fx, (hd[1] - cam.v0) / cam.fy]
    for hn in ([0.1, 0.2], [-0.3, 0.25], [0.0, 0.0]):
        assert np.allclose(cam.diff_distort(hn), fd(distort, hn), rtol=1e-6, atol=1e-8)


def test_unprojection_jacobian():                # cam.cpp:140-192
    cam = o.CamModel(o.Config.kinect(), T)
    for px in ([50.0, 60.0], [250.0, 180.0]):
        _, J = cam.unproject(px)
        Jn = fd(lambda x: cam.unproject(x)[0], px, eps=1e-4)
        assert np.allclose(J, Jn, rtol=1e-5, atol=1e-9)


def test_rotation_quaternion_partials():         # vR.cpp:1537-1566, 1654-1661
    rng = np.random.default_rng(2)
    q = rng.normal(size=4)
    d = rng.normal(size=3)
    J = o.jacobian_rq_d(q, d, T)
    Jn = fd(lambda x: o.quat2rot(x, T) @ d, q)
    assert np.allclose(J, Jn, rtol=1e-7, atol=1e-7)


def test_motion_jacobian():                      # vR.cpp:1492-1535 vs 1575-1589
    rng = np.random.default_rng(3)
    x = rng.normal(size=13) * 0.3
    x[3:7] = rng.normal(size=4); x[3:7] /= np.linalg.norm(x[3:7])
    dT = 0.05
    ctl = np.array([0.02, -0.01, 0.03])
    Ft = o.system_model_jacobian(x, dT, ctl, T)
    Jn = fd(lambda s: o.predict_state(s, [0.1, 0.2, 0.3], ctl, dT, T), x)
    assert np.allclose(Ft, Jn, rtol=1e-6, atol=1e-7)


def _filter_with_features(coding_xyz=False):
    cfg = o.Config.kinect()
    f = o.build_scenario(o.DenseFilter, cfg, 6, T)
    f.predict()
    z = o.synthetic_measurements(f, f.visible_indices())
    f.update(z)            # makes the state generic (camera moved away from anchors)
    if coding_xyz:
        for ft in f.features:
            pos = ft.position_in_state
            f.Sigma[pos + 5, pos + 5] = 1e-9      # force the linearity test to pass
        assert f.convert2xyz_if_linear_all() == 6
    return f


def _h_of_state(f, idx):
    def fun(mu):
        h, *_ = f.measure_feature(f.features[idx], mu=np.asarray(mu, dtype=T))
        return h
    return fun


def test_measurement_jacobian_inverse_depth():   # vR.cpp:508-551
    f = _filter_with_features()
    for idx in range(3):
        ft = f.features[idx]
        h, Hc, Hf, vis, rem = f.measure_feature(ft)
        Jn = fd(_h_of_state(f, idx), f.mu)
        pos = ft.position_in_state
        assert np.allclose(Hc, Jn[:, 0:7], rtol=1e-5, atol=1e-5)
        assert np.allclose(Hf, Jn[:, pos:pos + 6], rtol=1e-5, atol=1e-5)
        rest = np.delete(Jn, np.r_[0:7, pos:pos + 6], axis=1)
        assert np.allclose(rest, 0, atol=1e-7)


def test_measurement_jacobian_xyz():             # vR.cpp:552-578
    f = _filter_with_features(coding_xyz=True)
    for idx in range(3):
        ft = f.features[idx]
        assert ft.coding == o.XYZ
        h, Hc, Hf, vis, rem = f.measure_feature(ft)
        Jn = fd(_h_of_state(f, idx), f.mu)
        pos = ft.position_in_state
        assert Hf.shape == (2, 3)
        assert np.allclose(Hc, Jn[:, 0:7], rtol=1e-5, atol=1e-5)
        assert np.allclose(Hf, Jn[:, pos:pos + 3], rtol=1e-5, atol=1e-5)


def test_add_feature_jacobian():                 # vR.cpp:326-360, 1599-1623
    cfg = o.Config.kinect()
    f = o.DenseFilter(cfg, T)
    rng = np.random.default_rng(4)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    f.mu[0:3] = [0.2, -0.1, 0.4]
    f.mu[3:7] = q
    px = np.array([123.0, 77.0])
    fvec, G, Jp = f._add_feature_parts(*px)

    def new_feature(x):          # x = [r(3), q(4), u, v]
        g = o.DenseFilter(cfg, T)
        g.mu[0:7] = x[0:7]
        return g._add_feature_parts(x[7], x[8])[0]
    Jn = fd(new_feature, np.concatenate([f.mu[0:7], px]), eps=1e-5)
    assert np.allclose(G, Jn[:, 0:7], rtol=1e-5, atol=1e-7)
    assert np.allclose(Jp, Jn[:, 7:9], rtol=1e-4, atol=1e-9)


def test_inverse_depth_to_xyz_jacobian():        # vR.cpp:722-735
    f = o.DenseFilter(o.Config.kinect(), T)
    feat = np.array([0.1, -0.2, 0.3, 0.4, -0.3, 0.25])
    y, J, conv = f.inverse_depth_to_xyz_world(feat, 1)
    Jn = fd(lambda x: f.inverse_depth_to_xyz_world(x, 0)[0], feat)
    assert conv and np.allclose(J, Jn, rtol=1e-7, atol=1e-8)
