"""Symbolic cross-check of the analytic Jacobians: the forward functions of the hot path are
written down ONCE more, in sympy, differentiated symbolically, and the exact derivatives are compared with what the
oracle (and through it the HIP kernels) evaluate.  This is the build's own derivation -- not the reference's
notebook, which disagrees with its C++ in places -- and, unlike the finite-difference checks of
test_oracle_jacobians.py, it has no step-size error: agreement is to fp64 rounding.

Forward functions restated here (file:line of what they model):
  quat2rot                                vR.cpp:1408-1421
  projectAndDistort                       cam.cpp:68-111
  h(r, q, feature) inverse depth / XYZ    vR.cpp:508-578
  [theta, phi](hW) of addFeature          vR.cpp:326-346
  Predict_State's quaternion step         vR.cpp:1575-1589, 1388-1400
  inverseDepth2XyzWorld                   vR.cpp:690-738
"""
import numpy as np
import sympy as sp

import ekf_oracle as o

T = np.float64


def s_quat2rot(q):
    qr, qi, qj, qk = q
    return sp.Matrix([
        [qr*qr + qi*qi - qj*qj - qk*qk, -2*qr*qk + 2*qi*qj, 2*qr*qj + 2*qi*qk],
        [2*qr*qk + 2*qi*qj, qr*qr - qi*qi + qj*qj - qk*qk, -2*qr*qi + 2*qj*qk],
        [-2*qr*qj + 2*qi*qk, 2*qr*qi + 2*qj*qk, qr*qr - qi*qi - qj*qj + qk*qk]])


def s_project(cam, hC):
    x1, y1 = hC[0] / hC[2], hC[1] / hC[2]
    r2 = x1*x1 + y1*y1
    L = 1 + cam["k1"]*r2 + cam["k2"]*r2**2 + cam["k3"]*r2**3
    x2 = x1*L + 2*cam["p1"]*x1*y1 + cam["p2"]*(r2 + 2*x1*x1)
    y2 = y1*L + 2*cam["p2"]*x1*y1 + cam["p1"]*(r2 + 2*y1*y1)
    return sp.Matrix([cam["fx"]*x2 + cam["u0"], cam["fy"]*y2 + cam["v0"]])


def cam_values(cfg):
    c = o.CamModel(cfg, T)
    return {k: float(getattr(c, k)) for k in ("fx", "fy", "u0", "v0", "k1", "k2", "k3", "p1", "p2")}


def numeric(expr_matrix, symbols, values):
    f = sp.lambdify(symbols, expr_matrix, "numpy")
    return np.asarray(f(*values), dtype=T)


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def test_measurement_jacobian_inverse_depth_and_xyz():          # a4, a5, a6
    r = sp.symbols("r0:3")
    q = sp.symbols("q0:4")
    f = sp.symbols("f0:6")                                       # a (3), theta, phi, rho
    y = sp.symbols("y0:3")
    for cfg in (o.Config.kinect(), o.Config()):                  # mild and strong distortion
        cam = cam_values(cfg)
        qbar = (q[0], -q[1], -q[2], -q[3])
        Rcw = s_quat2rot(qbar)
        m = sp.Matrix([sp.sin(f[3])*sp.cos(f[4]), -sp.sin(f[4]), sp.cos(f[3])*sp.cos(f[4])])
        d_inv = f[5] * (sp.Matrix(f[0:3]) - sp.Matrix(r)) + m
        h_inv = s_project(cam, Rcw * d_inv)
        h_xyz = s_project(cam, Rcw * (sp.Matrix(y) - sp.Matrix(r)))
        J_inv_c = h_inv.jacobian(list(r) + list(q))
        J_inv_f = h_inv.jacobian(list(f))
        J_xyz_c = h_xyz.jacobian(list(r) + list(q))
        J_xyz_f = h_xyz.jacobian(list(y))
        rng = np.random.default_rng(11)
        for trial in range(4):
            flt = o.StructuredFilter(cfg, T)
            mu = flt.mu.copy()
            mu[0:3] = rng.normal(size=3) * 0.2
            qq = np.array([0.0, 0.0, -0.707106781, 0.707106781]) + rng.normal(size=4) * 0.05
            mu[3:7] = qq                                         # NOT re-normalised: the formulas hold off the sphere too
            Rwc = o.quat2rot(mu[3:7], T)
            hC = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), 1.0]) * rng.uniform(2, 8)
            yw = mu[0:3] + Rwc @ hC                              # a point in front of the camera
            rho = rng.uniform(0.1, 0.5)
            a = mu[0:3] + rng.normal(size=3) * 0.3
            mvec = (yw - a) * rho                                # y = a + m / rho  ->  m = rho (y - a): rescale to unit
            mvec /= np.linalg.norm(mvec)
            theta, phi = np.arctan2(mvec[0], mvec[2]), np.arctan2(-mvec[1], np.hypot(mvec[0], mvec[2]))
            rho = 1.0 / np.linalg.norm(yw - a)
            feat = np.concatenate([a, [theta, phi, rho]])
            flt.mu = np.concatenate([mu, feat, yw])
            fi = o.Feature(position_in_state=flt.camera_dim, coding=o.INV)
            fx = o.Feature(position_in_state=flt.camera_dim + 6, coding=o.XYZ)
            h1, Hc1, Hf1, _, _ = flt.measure_feature(fi)
            h2, Hc2, Hf2, _, _ = flt.measure_feature(fx)
            vals = list(mu[0:3]) + list(mu[3:7])
            assert rel(h1, numeric(h_inv, list(r) + list(q) + list(f), vals + list(feat)).ravel()) < 1e-13
            assert rel(Hc1, numeric(J_inv_c, list(r) + list(q) + list(f), vals + list(feat))) < 1e-11
            assert rel(Hf1, numeric(J_inv_f, list(r) + list(q) + list(f), vals + list(feat))) < 1e-11
            assert rel(h2, numeric(h_xyz, list(r) + list(q) + list(y), vals + list(yw)).ravel()) < 1e-13
            assert rel(Hc2, numeric(J_xyz_c, list(r) + list(q) + list(y), vals + list(yw))) < 1e-11
            assert rel(Hf2, numeric(J_xyz_f, list(r) + list(q) + list(y), vals + list(yw))) < 1e-11
            assert np.allclose(h1, h2, atol=1e-9)                # same world point, two parametrisations


def test_distortion_matrix_is_the_exact_derivative():          # cam.cpp:18-47 vs :78-92
    x1, y1 = sp.symbols("x1 y1")
    for cfg in (o.Config.kinect(), o.Config()):
        cam = cam_values(cfg)
        unit = dict(cam, fx=1.0, fy=1.0, u0=0.0, v0=0.0)
        D = s_project(unit, sp.Matrix([x1, y1, 1])).jacobian([x1, y1])
        c = o.CamModel(cfg, T)
        for hn in ([0.1, 0.2], [-0.3, 0.25], [0.0, 0.0], [0.45, -0.35]):
            assert rel(c.diff_distort(hn), numeric(D, [x1, y1], hn)) < 1e-13


def test_add_feature_angle_partials():                         # vR.cpp:1599-1623 (rows theta, phi)
    hx, hy, hz = sp.symbols("hx hy hz")
    ang = sp.Matrix([sp.atan2(hx, hz), sp.atan2(-hy, sp.sqrt(hx*hx + hz*hz))])
    J = ang.jacobian([hx, hy, hz])
    for hW in ([0.3, -0.2, 2.0], [-1.0, 0.7, 0.4], [0.05, 0.9, -1.5]):
        assert rel(o.jacobian_inv_feature_to_hW(hW, T)[3:5], numeric(J, [hx, hy, hz], hW)) < 1e-13


def test_motion_jacobian_quaternion_block():                   # a2: vR.cpp:1492-1535 against a3: :1575-1589
    q = sp.symbols("q0:4")
    w = sp.symbols("w0:3")
    dT = sp.Symbol("dT")
    nw = sp.sqrt(w[0]**2 + w[1]**2 + w[2]**2)
    alpha = dT * nw                                              # vec2quat(dT w): angle alpha, axis w / |w|
    h = sp.Matrix([sp.cos(alpha / 2)] + [dT * w[i] * sp.sin(alpha / 2) / alpha for i in range(3)])
    Ups = sp.Matrix([[q[0], -q[1], -q[2], -q[3]], [q[1], q[0], -q[3], q[2]],
                     [q[2], q[3], q[0], -q[1]], [q[3], -q[2], q[1], q[0]]])
    qn = Ups * h
    Jq, Jw = qn.jacobian(list(q)), qn.jacobian(list(w))
    rng = np.random.default_rng(5)
    for trial in range(4):
        x = rng.normal(size=13) * 0.3
        x[3:7] = rng.normal(size=4)
        x[3:7] /= np.linalg.norm(x[3:7])
        dt = 1.0 / 30.0
        Ft = o.system_model_jacobian(x, dt, (0, 0, 0), T)
        vals = list(x[3:7]) + list(x[10:13]) + [dt]
        assert rel(Ft[3:7, 3:7], numeric(Jq, list(q) + list(w) + [dT], vals)) < 1e-12
        assert rel(Ft[3:7, 10:13], numeric(Jw, list(q) + list(w) + [dT], vals)) < 1e-10
        assert np.array_equal(Ft[0:3, 7:10], dt * np.eye(3)) and np.array_equal(Ft[7:13, 7:13], np.eye(6))


def test_inverse_depth_to_xyz_jacobian():                      # a14: vR.cpp:690-738
    f = sp.symbols("f0:6")
    m = sp.Matrix([sp.sin(f[3])*sp.cos(f[4]), -sp.sin(f[4]), sp.cos(f[3])*sp.cos(f[4])])
    yv = sp.Matrix(f[0:3]) + m / f[5]
    J = yv.jacobian(list(f))
    flt = o.StructuredFilter(o.Config.kinect(), T)
    for feat in ([0.1, -0.2, 0.3, 0.4, -0.1, 0.25], [1.0, 2.0, -0.5, -1.2, 0.6, 0.08]):
        y, Jy, _ = flt.inverse_depth_to_xyz_world(np.array(feat), 1)
        assert rel(y, numeric(yv, list(f), feat).ravel()) < 1e-14
        assert rel(Jy, numeric(J, list(f), feat)) < 1e-13
