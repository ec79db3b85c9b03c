"""CPU oracle for the EKF-MonoSLAM predict/update core.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference's filter arithmetic
(`mono-slam/src/vslamRansac.cpp`, `mono-slam/src/camModel.cpp` of
engyasin/EKF-MonoSLAM_for_3D-reconstruction).  It is *the checker*, never the
product: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import it.  The shipped path is the HIP library behind
`include/ekf_monoslam.h` and fails loudly when that library is missing.

PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for
this path, and it cannot be compiled here (Eigen3 / OpenCV / ROS / libconfig++
are absent).  What pins this oracle instead (tests/test_oracle_*.py):
known-answer cases derivable from the source, finite-difference checks of every
analytic Jacobian in fp64, the add-feature -> predict-measurement round trip,
and agreement of the two flavours below on the same inputs.

Two flavours, same public methods:

* `DenseFilter`  -- faithful-dense: every covariance transform is the same
  dense n x n triple product, in the same order, as the reference
  (vR.cpp:367, 477, 598, 765, 1268-1280, 1641), with GEMM / inverse delegated
  to numpy (OpenBLAS) in the working dtype.
* `StructuredFilter` -- the same arithmetic with the identity blocks of the
  Jacobians exploited (strips / compact H).  Used where dense is too slow.

Citations "vR.cpp:N" are lines of mono-slam/src/vslamRansac.cpp, "cam.cpp:N"
of mono-slam/src/camModel.cpp, "cfg.cpp:N" of mono-slam/src/ConfigVSLAM.cpp.
All matrices here are mathematical (row, col) numpy arrays; the reference is
fp32 throughout (vR.hpp:32-44,67-68) except `dT` (double, vR.hpp:30).
"""
from __future__ import annotations

import dataclasses
import math
from typing import List, Optional

import numpy as np

STATE_DIM = 14  # vR.cpp:22  (r3, q4, v3, w3, map_scale)
INV, XYZ = 0, 1  # Patch.hpp:9-10 (coding)


# --------------------------------------------------------------------------
# configuration (cfg.cpp:27-47 defaults; conf/*.cfg parameter sets)
# --------------------------------------------------------------------------
@dataclasses.dataclass
class Config:
    sigma_vx: float = 0.01
    sigma_vy: float = 0.01
    sigma_vz: float = 0.01
    sigma_wx: float = 0.01
    sigma_wy: float = 0.01
    sigma_wz: float = 0.01
    rho_0: float = 0.1
    sigma_rho_0: float = 0.25
    window_size: int = 21
    sigma_pixel: int = 2
    kernel_size: int = 0
    sigma_size: int = 2
    scale: int = 1
    T_camera: float = 0.5
    nInitFeatures: int = 5
    min_features: int = 30
    max_features: int = 100
    forsePlane: int = 0
    # camConfig (camModel.hpp:9-11); already divided by `scale` (cfg.cpp:87-103)
    fx: float = 592.2860
    fy: float = 584.9968
    u0: float = 362.1059
    v0: float = 275.9642
    k1: float = -0.3954
    k2: float = 0.5521
    k3: float = 0.0
    p1: float = -0.0075
    p2: float = 0.0140
    # frame.size() after captureNewFrame's resize (vR.cpp:236)
    image_width: int = 640
    image_height: int = 480

    @staticmethod
    def kinect() -> "Config":
        """mono-slam/conf/conf_kinect.cfg, scale = 2 (SURVEY 8d primary set)."""
        s = 2.0
        return Config(
            sigma_vx=0.03, sigma_vy=0.03, sigma_vz=0.03,
            sigma_wx=0.015, sigma_wy=0.015, sigma_wz=0.015,
            rho_0=0.2, sigma_rho_0=0.25, sigma_pixel=2, window_size=15,
            kernel_size=1000, scale=2, T_camera=0.0,
            nInitFeatures=50, min_features=100, max_features=1000,
            fx=537.673722507338 / s, fy=534.380205679756 / s,
            u0=321.226061527052 / s, v0=249.773992466202 / s,
            k1=0.0395956005042652, k2=-0.111310452999064, k3=0.0,
            p1=0.00211988964071199, p2=0.00070924348636878,
            image_width=320, image_height=240)

    @staticmethod
    def sim() -> "Config":
        """conf_sim.cfg, scale = 10, zero distortion (second parameter set)."""
        s = 10.0
        return Config(
            sigma_vx=0.0000008, sigma_vy=0.0000008, sigma_vz=0.00000000008,
            sigma_wx=0.000000004, sigma_wy=0.0000004, sigma_wz=0.00000000004,
            rho_0=0.1, sigma_rho_0=0.25, sigma_pixel=2, window_size=30,
            kernel_size=3, scale=10, T_camera=0.2, sigma_size=4,
            min_features=20, max_features=35,
            fx=2217.0187 / s, fy=2217.0187 / s, u0=1280.5 / s, v0=960.5 / s,
            k1=0.0, k2=0.0, k3=0.0, p1=0.0, p2=0.0,
            image_width=256, image_height=192)


# --------------------------------------------------------------------------
# quaternion / rotation helpers (vR.cpp:1388-1460, 1537-1595)
# --------------------------------------------------------------------------
def vec2quat(vec, T):
    """vR.cpp:1388-1400."""
    vec = np.asarray(vec, dtype=T)
    alpha = T(np.sqrt(np.dot(vec, vec)))
    if alpha != 0:
        q = np.empty(4, dtype=T)
        q[0] = T(np.cos(alpha / T(2)))
        q[1:4] = vec * T(np.sin(alpha / T(2))) / alpha
        return q
    return np.array([1, 0, 0, 0], dtype=T)  # quatZero, vR.cpp:1402-1406


def quat2rot(q, T):
    """vR.cpp:1408-1421."""
    qr, qi, qj, qk = (T(x) for x in q)
    two = T(2)
    return np.array([
        [qr*qr + qi*qi - qj*qj - qk*qk, -two*qr*qk + two*qi*qj, two*qr*qj + two*qi*qk],
        [two*qr*qk + two*qi*qj, qr*qr - qi*qi + qj*qj - qk*qk, -two*qr*qi + two*qj*qk],
        [-two*qr*qj + two*qi*qk, two*qr*qi + two*qj*qk, qr*qr - qi*qi - qj*qj + qk*qk],
    ], dtype=T)


def upsilon(q, T):
    """YupsilonMatric, vR.cpp:1423-1438."""
    r, x, y, z = (T(v) for v in q)
    return np.array([[r, -x, -y, -z],
                     [x, r, -z, y],
                     [y, z, r, -x],
                     [z, -y, x, r]], dtype=T)


def upsilon_bar(q, T):
    """YupsilonMatricComplementar, vR.cpp:1441-1455."""
    r, x, y, z = (T(v) for v in q)
    return np.array([[r, -x, -y, -z],
                     [x, r, z, -y],
                     [y, -z, r, x],
                     [z, y, -x, r]], dtype=T)


def quat_product(q1, q2, T):
    """quatCrossProduct, vR.cpp:1457-1460."""
    return upsilon(q1, T) @ np.asarray(q2, dtype=T)


def quat_complement(q, T):
    """vR.cpp:1568-1572."""
    q = np.asarray(q, dtype=T)
    return np.array([q[0], -q[1], -q[2], -q[3]], dtype=T)


def diff_quat2rot(q, index, T):
    """dR(q)/dq_index, vR.cpp:1537-1566."""
    q0, qx, qy, qz = (T(2) * T(v) for v in q)
    if index == 0:
        m = [[q0, -qz, qy], [qz, q0, -qx], [-qy, qx, q0]]
    elif index == 1:
        m = [[qx, qy, qz], [qy, -qx, -q0], [qz, q0, -qx]]
    elif index == 2:
        m = [[-qy, qx, q0], [qx, qy, qz], [-q0, qz, -qy]]
    else:
        m = [[-qz, -q0, qx], [q0, -qz, qy], [qx, qy, qz]]
    return np.array(m, dtype=T)


def jacobian_rq_d(q, d, T):
    """Jacobian_hW_to_qantrion: d(R(q) d)/dq, 3x4, vR.cpp:1654-1661."""
    d = np.asarray(d, dtype=T)
    out = np.empty((3, 4), dtype=T)
    for j in range(4):
        out[:, j] = diff_quat2rot(q, j, T) @ d
    return out


def d_qbar_q(T):
    """vR.cpp:1591-1595."""
    return np.diag(np.array([1, -1, -1, -1], dtype=T))


# --------------------------------------------------------------------------
# motion model (vR.cpp:1492-1535, 1575-1589)
# --------------------------------------------------------------------------
def jacobian_qt_w(q, w, dT, T):
    """Jacobian_qt_w, vR.cpp:1516-1535.  `dT` is double in the reference, so the
    trigonometric arguments are formed in double and rounded to T afterwards.
    Quirk fixed: the reference leaves n_w uninitialised when |w|=0
    (it is multiplied by s=0 / (c-Sinc)=0 there); n_w := 0 here."""
    w = np.asarray(w, dtype=T)
    n = T(np.sqrt(np.dot(w, w)))
    s = T(math.sin(float(dT) * float(n) / 2.0))
    c = T(math.cos(float(dT) * float(n) / 2.0))
    sinc = T(1) if n == 0 else T(2.0 * math.sin(float(dT) * float(n) / 2.0) / (float(dT) * float(n)))
    n_w = w / n if n > 0 else np.zeros(3, dtype=T)
    t2 = np.zeros((4, 3), dtype=T)
    t2[0, :] = T(-float(dT) * 0.5 * float(s)) * n_w
    t2[1:4, :] = T(float(dT) * 0.5) * (sinc * np.eye(3, dtype=T) + (c - sinc) * np.outer(n_w, n_w))
    return upsilon(q, T) @ t2


def system_model_jacobian(x13, dT, r_ctl, T):
    """System_model_jacobian, vR.cpp:1492-1507.  Returns Ft (13x13)."""
    x13 = np.asarray(x13, dtype=T)
    q = x13[3:7]
    w = x13[10:13]
    r_ctl = np.asarray(r_ctl, dtype=T)
    h = vec2quat(T(dT) * (w + r_ctl), T)
    Ft = np.eye(13, dtype=T)
    Ft[3:7, 3:7] = upsilon_bar(h, T)                       # Jacobian_qt_qt1, vR.cpp:1510
    Ft[3:7, 10:13] = jacobian_qt_w(q, w + r_ctl, dT, T)
    Ft[0:3, 7:10] = T(dT) * np.eye(3, dtype=T)
    return Ft


def predict_state(x13, t_ctl, r_ctl, dT, T):
    """Predict_State, vR.cpp:1575-1589 (controls persist in v, w)."""
    x = np.array(x13, dtype=T)
    v = x[7:10] + np.asarray(t_ctl, dtype=T)
    w = x[10:13] + np.asarray(r_ctl, dtype=T)
    x[0:3] = x[0:3] + T(dT) * v
    x[3:7] = quat_product(x[3:7], vec2quat(T(dT) * w, T), T)
    x[7:10] = v
    x[10:13] = w
    return x


# --------------------------------------------------------------------------
# camera model (cam.cpp:18-192)
# --------------------------------------------------------------------------
class CamModel:
    def __init__(self, cfg: Config, T):
        self.T = T
        for k in ("fx", "fy", "u0", "v0", "k1", "k2", "k3", "p1", "p2"):
            setattr(self, k, T(np.float32(getattr(cfg, k))))   # camConfig fields are float

    def diff_distort(self, hn):
        """diff_distort_undistort, cam.cpp:18-47: D(hn), 2x2."""
        T = self.T
        x, y = T(hn[0]), T(hn[1])
        r2 = x*x + y*y
        L = T(1) + self.k1*r2 + self.k2*r2*r2 + self.k3*r2*r2*r2
        f = self.k1 + T(2)*self.k2*r2 + T(3)*self.k3*r2*r2
        hn = np.array([x, y], dtype=T)
        hn_c = np.array([y, x], dtype=T)
        pv = np.array([self.p1, self.p2], dtype=T)
        pv_c = np.array([self.p2, self.p1], dtype=T)
        jm = np.array([[self.p2*x, 0], [0, self.p1*y]], dtype=T)
        return (L*np.eye(2, dtype=T) + T(2)*f*np.outer(hn, hn) + T(2)*np.outer(pv, hn_c)
                + T(2)*np.outer(pv_c, hn) + T(4)*jm)

    def project(self, hC, want_J=True):
        """projectAndDistort, cam.cpp:68-138.  Returns (hd[2], J[2x3] or None)."""
        T = self.T
        x, y, z = T(hC[0]), T(hC[1]), T(hC[2])
        x1 = x / z
        y1 = y / z
        r2 = x1*x1 + y1*y1
        l = T(1) + self.k1*r2 + self.k2*r2*r2 + self.k3*r2*r2*r2
        x2 = x1*l + T(2)*self.p1*x1*y1 + self.p2*(r2 + T(2)*x1*x1)
        y2 = y1*l + T(2)*self.p2*x1*y1 + self.p1*(r2 + T(2)*y1*y1)
        hd = np.array([self.fx*x2 + self.u0, self.fy*y2 + self.v0], dtype=T)
        if not want_J:
            return hd, None
        Jn = np.array([[T(1)/z, 0, -x/z/z], [0, T(1)/z, -y/z/z]], dtype=T)
        Jp = np.array([[self.fx, 0], [0, self.fy]], dtype=T)
        J = Jp @ self.diff_distort((x1, y1)) @ Jn
        return hd, J

    def unproject(self, hd):
        """UndistortAndDeproject, cam.cpp:140-192.  Returns (hC[3], J[3x2])."""
        T = self.T
        x2 = (T(hd[0]) - self.u0) / self.fx
        y2 = (T(hd[1]) - self.v0) / self.fy
        x1, y1 = x2, y2
        for _ in range(50):  # cam.cpp:162-174
            r2 = x1*x1 + y1*y1
            l = T(1) + self.k1*r2 + self.k2*r2*r2 + self.k3*r2*r2*r2
            dx = T(2)*self.p1*x1*y1 + self.p2*(r2 + T(2)*x1*x1)
            dy = T(2)*self.p2*x1*y1 + self.p1*(r2 + T(2)*y1*y1)
            x1 = (x2 - dx) / l
            y1 = (y2 - dy) / l
        hC = np.array([x1, y1, 1], dtype=T)
        D = self.diff_distort((x1, y1))
        det = D[0, 0]*D[1, 1] - D[0, 1]*D[1, 0]
        Dinv = np.array([[D[1, 1], -D[0, 1]], [-D[1, 0], D[0, 0]]], dtype=T) / det
        Ju = np.array([[1, 0], [0, 1], [0, 0]], dtype=T)
        Jk = np.array([[T(1)/self.fx, 0], [0, T(1)/self.fy]], dtype=T)
        return hC, Ju @ Dinv @ Jk


# --------------------------------------------------------------------------
# feature parametrisation helpers
# --------------------------------------------------------------------------
def m_vec(theta, phi, T):
    """vR.cpp:1468-1471."""
    theta, phi = T(theta), T(phi)
    return np.array([np.sin(theta)*np.cos(phi), -np.sin(phi), np.cos(theta)*np.cos(phi)], dtype=T)


def inverse2xyz_projecting(f, r, T, want_J=True):
    """inverse2XYZ4_projecting, vR.cpp:1462-1489: d = rho (a - r) + m, J 3x6."""
    f = np.asarray(f, dtype=T)
    r = np.asarray(r, dtype=T)
    theta, phi, ro = f[3], f[4], f[5]
    m = m_vec(theta, phi, T)
    J = None
    if want_J:
        J = np.empty((3, 6), dtype=T)
        J[:, 0:3] = ro * np.eye(3, dtype=T)
        J[:, 3] = [np.cos(theta)*np.cos(phi), 0, -np.sin(theta)*np.cos(phi)]
        J[:, 4] = [-np.sin(theta)*np.sin(phi), -np.cos(phi), -np.cos(theta)*np.sin(phi)]
        J[:, 5] = f[0:3] - r
    return ro * (f[0:3] - r) + m, J


def jacobian_inv_feature_to_hW(hW, T):
    """Jacobain_inv_feature_to_hW, vR.cpp:1599-1623: d[a,theta,phi,rho]/dhW, 6x3."""
    hx, hy, hz = (T(v) for v in hW)
    normal = hx*hx + hz*hz
    normal2 = hx*hx + hy*hy + hz*hz
    J = np.zeros((6, 3), dtype=T)
    J[3, :] = [hz/normal, 0, -hx/normal]
    sn = T(np.sqrt(normal))
    J[4, :] = [hx*hy/sn/normal2, -sn/normal2, hz*hy/sn/normal2]
    return J


def is_inside_image(hi, width, height, window_size):
    """isInsideImage, vR.cpp:1644-1652 (integer window/2)."""
    half = int(window_size) // 2
    i, j = float(hi[0]), float(hi[1])
    return (i > half) and (j > half) and (i < width - half) and (j < height - half)


@dataclasses.dataclass
class Feature:
    """The hot-path fields of `Patch` (Patch.hpp:20-28, 35, 64-65, 86-92)."""
    position_in_state: int
    coding: int = INV
    position_in_z: int = -1
    is_in_innovation: bool = False
    remove_flag: bool = False
    h: Optional[np.ndarray] = None   # 2
    Hc: Optional[np.ndarray] = None  # 2x7   (compact: columns 0..6 of the dense 2xn row pair)
    Hf: Optional[np.ndarray] = None  # 2x6 or 2x3 (columns pos..pos+size)
    z: Optional[np.ndarray] = None
    real_index: int = 0              # Patch::real_index = patchnumbre at creation (Patch.cpp:84, vR.cpp:318-319)
    n_find: int = 1                  # Patch::n_find (Patch.cpp:86), +1 per update in which the patch is in Li or Hi (:145)

    @property
    def size(self) -> int:
        return 3 if self.coding == XYZ else 6


# --------------------------------------------------------------------------
# faithful-dense filter
# --------------------------------------------------------------------------
class DenseFilter:
    """Restates `VSlamFilter`'s math methods with the reference's dense products."""

    structured = False

    def __init__(self, cfg: Config, dtype=np.float32, camera_dim: int = STATE_DIM):
        assert camera_dim in (13, 14)
        self.cfg = cfg
        self.T = T = np.dtype(dtype).type
        self.camera_dim = camera_dim
        self.cam = CamModel(cfg, T)
        self.dT = 1.0                                           # vR.cpp:155
        self.sigma_pixel_2 = int(cfg.sigma_pixel) * int(cfg.sigma_pixel)  # vR.cpp:150-151 (ints)
        # vR.cpp:163-180
        mu = np.zeros(camera_dim, dtype=T)
        mu[3:7] = [0.0, 0.0, -0.707106781, 0.707106781]
        if camera_dim == 14:
            mu[13] = 1
        self.mu = mu
        # vR.cpp:194-202
        Vmax = np.eye(6, dtype=T)
        for i, k in enumerate(("sigma_vx", "sigma_vy", "sigma_vz", "sigma_wx", "sigma_wy", "sigma_wz")):
            s = T(np.float32(getattr(cfg, k)))                 # ConfigVSLAM fields are float
            Vmax[i, i] = s * s
        self.Vmax = Vmax
        self.Vmax_n = Vmax * T(2)
        # vR.cpp:146, 211-216
        S = T(0.0000000004) * np.eye(camera_dim, dtype=T)
        if camera_dim == 14:
            S[13, 13] = T(0.09)
        sv = T(0.0004)
        S[7:10, 7:10] = sv * sv * np.eye(3, dtype=T)
        S[10:13, 10:13] = sv * sv * np.eye(3, dtype=T)
        self.Sigma = S
        self.features: List[Feature] = []
        self.patchnumbre = 1                                    # vR.cpp:148
        self.deleted_patches = []                               # (real_index, XYZ_pos (3), cov_4_delete (9)), vR.cpp:394-404
        self.St = None
        self.Kt = None
        self.Ft = None
        self.h_out = None
        self.map_scale = T(1)

    # ---- getters (vR.cpp:127-140, 247, 841-866) ---------------------------
    def num_features(self):
        return len(self.features)

    @property
    def n(self):
        return self.mu.shape[0]

    def get_state(self):
        return self.mu[:self.camera_dim].copy()

    def get_sigma(self):
        return self.Sigma[:self.camera_dim, :self.camera_dim].copy()

    def covariance_parameter(self):
        d = np.diag(self.Sigma)
        return self.T(d[0] + d[1] + d[2] + d[4] + d[5] + d[6] + d[3])

    # ---- add feature (vR.cpp:309-371) --------------------------------------
    def _add_feature_parts(self, u, v):
        """Shared by both flavours: new 6-vector, G (6x7: d f/d[r,q]), Jp (6x2), pixel test."""
        T = self.T
        cfg = self.cfg
        hd = np.array([u, v], dtype=T)
        if not is_inside_image(hd, cfg.image_width, cfg.image_height, cfg.window_size):
            return None
        r = self.mu[0:3]
        q = self.mu[3:7]
        hC, J_undist = self.cam.unproject(hd)
        Rot = quat2rot(q, T)
        hW = Rot @ hC
        hx, hy, hz = hW
        theta = T(np.arctan2(hx, hz))
        phi = T(np.arctan2(-hy, np.sqrt(hx*hx + hz*hz)))
        f = np.array([r[0], r[1], r[2], theta, phi, T(np.float32(cfg.rho_0))], dtype=T)
        J_f_hW = jacobian_inv_feature_to_hW(hW, T)
        J_hW_q = jacobian_rq_d(q, hC, T)
        G = np.zeros((6, 7), dtype=T)
        G[0:3, 0:3] = np.eye(3, dtype=T)          # Js.block<3,3>(nOld,0), vR.cpp:352
        G[:, 3:7] = J_f_hW @ J_hW_q               # vR.cpp:358
        Jp = J_f_hW @ Rot @ J_undist              # vR.cpp:359
        return f, G, Jp

    def _new_feature(self, pos):
        """Patch(..., pos, patchnumbre); patchnumbre += 1 (vR.cpp:318-319)."""
        ft = Feature(position_in_state=pos, coding=INV, real_index=self.patchnumbre)
        self.patchnumbre += 1
        return ft

    def add_feature(self, u, v) -> int:
        T = self.T
        parts = self._add_feature_parts(u, v)
        if parts is None:
            return 0
        f, G, Jp = parts
        nOld = self.n
        self.features.append(self._new_feature(nOld))
        self.mu = np.concatenate([self.mu, f])
        Js = np.zeros((nOld + 6, nOld + 3), dtype=T)
        Js[:nOld, :nOld] = np.eye(nOld, dtype=T)
        Js[nOld:nOld+6, 0:7] = G
        Js[nOld:nOld+6, nOld:nOld+2] = Jp
        Js[nOld+5, nOld+2] = 1
        S = T(self.sigma_pixel_2) * np.eye(nOld + 3, dtype=T)
        S[:nOld, :nOld] = self.Sigma
        S[nOld+2, nOld+2] = T(np.float32(self.cfg.sigma_rho_0))     # unsquared, vR.cpp:365
        self.Sigma = Js @ S @ Js.T                       # vR.cpp:367
        return 1

    # ---- remove feature (vR.cpp:373-421) -----------------------------------
    def remove_feature(self, index):
        p = self.features[index]
        pos, psize = p.position_in_state, p.size
        if p.n_find > 5 and psize == 3:                                   # "a segment to save good features", vR.cpp:394-404
            # XYZ_pos = inverseDepth2XyzWorld(3-vector) = the vector itself (:699-700); cov_4_delete(i) = the transposed
            # 3x3 block in Eigen's column-major linear order = the block row by row
            self.deleted_patches.append((p.real_index, self.mu[pos:pos+3].copy(),
                                         self.Sigma[pos:pos+3, pos:pos+3].reshape(-1).copy()))
        keep = np.r_[0:pos, pos+psize:self.n]
        self.mu = self.mu[keep]
        self.Sigma = self.Sigma[np.ix_(keep, keep)]
        for ft in self.features[index+1:]:
            ft.position_in_state -= psize
        del self.features[index]

    # ---- predict (vR.cpp:451-603) ------------------------------------------
    def _motion(self, t_ctl, r_ctl, vcontrol):
        T = self.T
        Ft = system_model_jacobian(self.mu[0:13], self.dT, r_ctl, T)
        V = self.Vmax if vcontrol else self.Vmax_n                       # vR.cpp:463-473
        Vs = V / T(self.dT) / T(self.dT)
        Q = Ft[:, 7:13] @ Vs @ Ft[:, 7:13].T
        return Ft, Q

    def predict_covariance(self, Ft, Q):
        T = self.T
        n = self.n
        F = np.eye(n, dtype=T)
        F[0:13, 0:13] = Ft
        Qtot = np.zeros((n, n), dtype=T)
        Qtot[0:13, 0:13] = Q
        self.Sigma = F @ self.Sigma @ F.T + Qtot                          # vR.cpp:477

    def predict(self, t_ctl=(0, 0, 0), r_ctl=(0, 0, 0), vcontrol=False):
        T = self.T
        Ft, Q = self._motion(t_ctl, r_ctl, vcontrol)
        self.Ft = Ft
        self.Q = Q
        self.predict_covariance(Ft, Q)
        self.mu[0:13] = predict_state(self.mu[0:13], t_ctl, r_ctl, self.dT, T)   # vR.cpp:480
        if self.camera_dim == 14:
            self.map_scale = self.mu[13]
        self.measure()
        self.St = self.innovation_covariance(self.visible_indices())
        return self

    def measure_feature(self, ft: Feature, mu=None, r=None, q=None):
        """One iteration of the loop at vR.cpp:508-579 (image blur excluded).
        Returns (h, Hc 2x7, Hf 2xsize, visible, remove)."""
        T = self.T
        cfg = self.cfg
        mu = self.mu if mu is None else mu
        r = mu[0:3] if r is None else r
        q = mu[3:7] if q is None else q
        qc = quat_complement(q, T)
        RotCW = quat2rot(qc, T)
        pos = ft.position_in_state
        if ft.coding == INV:
            f = mu[pos:pos+6]
            rem = bool(f[5] <= 0)                                       # vR.cpp:517-522
            d, J_hW_f = inverse2xyz_projecting(f, r, T, True)
            scale_r = -f[5]
        else:
            y = mu[pos:pos+3]
            d = y - r
            J_hW_f = np.eye(3, dtype=T)
            scale_r = T(-1)
            rem = False
        hC = RotCW @ d
        hi, J_h_hC = self.cam.project(hC, True)
        vis = (not rem) and is_inside_image(hi, cfg.image_width, cfg.image_height, cfg.window_size) \
            and bool(hC[2] >= 0)
        J_hC_q = jacobian_rq_d(qc, d, T) @ d_qbar_q(T)                   # vR.cpp:537
        Hc = np.zeros((2, 7), dtype=T)
        Hc[:, 0:3] = scale_r * (J_h_hC @ RotCW)                          # vR.cpp:539 / 568
        Hc[:, 3:7] = J_h_hC @ J_hC_q                                     # vR.cpp:540
        Hf = J_h_hC @ RotCW @ J_hW_f                                     # vR.cpp:541 / 570
        return hi, Hc, Hf, vis, rem

    def measure(self):
        for ft in self.features:
            hi, Hc, Hf, vis, rem = self.measure_feature(ft)
            if rem:
                ft.remove_flag = True
            ft.is_in_innovation = vis
            # the reference `continue`s before storing h/H of an invisible or
            # rho <= 0 feature (vR.cpp:521, 533); the values are kept here so a
            # caller may still force such a feature into an update list.
            ft.h, ft.Hc, ft.Hf = hi, Hc, Hf
        j = 0
        for ft in self.features:                                        # vR.cpp:584-592
            if ft.is_in_innovation:
                ft.position_in_z = 2 * j
                j += 1
        vis = self.visible_indices()
        self.h_out = (np.concatenate([self.features[i].h for i in vis]) if vis
                      else np.zeros(0, dtype=self.T))

    def visible_indices(self):
        return [i for i, ft in enumerate(self.features) if ft.is_in_innovation]

    def dense_H(self, indices, plane=False):
        """Stack the zero-filled 2 x n row pairs (vR.cpp:510, 588; utils.cpp:19-27)."""
        T = self.T
        n = self.n
        rows = 2 * len(indices) + (3 if plane else 0)
        H = np.zeros((rows, n), dtype=T)
        for k, i in enumerate(indices):
            ft = self.features[i]
            H[2*k:2*k+2, 0:7] = ft.Hc
            H[2*k:2*k+2, ft.position_in_state:ft.position_in_state+ft.size] = ft.Hf
        if plane:                                                       # vR.cpp:1252-1255
            H[rows-3, 1] = 1
            H[rows-2, 4] = 1
            H[rows-1, 6] = 1
        return H

    def stacked_h(self, indices):
        if not indices:
            return np.zeros(0, dtype=self.T)
        return np.concatenate([self.features[i].h for i in indices])

    def innovation_covariance(self, indices, plane=False):
        """St = H Sigma H^T + R, vR.cpp:598 / 1268-1274."""
        T = self.T
        if not indices and not plane:
            return np.zeros((0, 0), dtype=T)
        H = self.dense_H(indices, plane)
        p = H.shape[0]
        R = T(self.sigma_pixel_2) * np.eye(p, dtype=T)
        if plane:
            R[p-3:, p-3:] = T(0.00001) * np.eye(3, dtype=T)              # vR.cpp:1246,1272
        return H @ self.Sigma @ H.T + R

    # ---- EKF update block (vR.cpp:1245-1284; same block at 1053-1061, 663-676)
    def update(self, z, indices=None, plane=None):
        """z: stacked pixel measurements (2 per listed feature, in list order).
        indices: feature indices measured (default: all visible).  The 1-point
        RANSAC / matching stages (vR.cpp:875-1130) are outside this block."""
        T = self.T
        if indices is None:
            indices = self.visible_indices()
        indices = list(indices)
        plane = bool(self.cfg.forsePlane) if plane is None else bool(plane)
        z = np.asarray(z, dtype=T).reshape(-1)
        h = self.stacked_h(indices)
        if plane:
            h = np.concatenate([h, np.array([self.mu[1], self.mu[4], self.mu[6]], dtype=T)])
            z = np.concatenate([z, np.zeros(3, dtype=T)])
        for k, i in enumerate(indices):
            self.features[i].position_in_z = 2 * k
            self.features[i].z = z[2*k:2*k+2].copy()
        if z.shape[0] > 0:
            self._update_block(indices, plane, z, h)
            self.normalize_quaternion()
        return self

    def _update_block(self, indices, plane, z, h):
        T = self.T
        H = self.dense_H(indices, plane)
        St = self.innovation_covariance(indices, plane)
        Kt = self.Sigma @ H.T @ np.linalg.inv(St)                        # vR.cpp:1276
        self.mu = self.mu + Kt @ (z - h)                                  # vR.cpp:1278
        n = self.n
        self.Sigma = (np.eye(n, dtype=T) - Kt @ H) @ self.Sigma           # vR.cpp:1279
        self.St, self.Kt = St, Kt

    def normalize_quaternion(self):
        """normalizeQuaternion, vR.cpp:1625-1642."""
        T = self.T
        q = self.mu[3:7].copy()
        norma = T(np.sqrt(np.dot(q, q)))
        self.mu[3:7] = q / norma
        Q = norma*norma*np.eye(4, dtype=T) - np.outer(q, q)
        Q = Q * (T(1) / (norma*norma*norma))
        self._apply_quat_block(Q)

    def _apply_quat_block(self, Q):
        T = self.T
        n = self.n
        Qc = np.eye(n, dtype=T)
        Qc[3:7, 3:7] = Q
        self.Sigma = Qc @ self.Sigma @ Qc.T                               # vR.cpp:1641

    # ---- inverse depth -> XYZ (vR.cpp:690-772) -----------------------------
    def inverse_depth_to_xyz_world(self, f, mode=0, pos=None):
        """inverseDepth2XyzWorld, vR.cpp:690-738.  mode 0: no J; 1: J; 2: decide by
        the linearity index (then `pos` is the feature's state offset).
        Returns (y, J or None, converted flag)."""
        T = self.T
        f = np.asarray(f, dtype=T)
        if f.shape[0] == 3:
            return f.copy(), None, False
        theta, phi, ro = f[3], f[4], f[5]
        m = m_vec(theta, phi, T)
        y = f[0:3] + m / ro
        want = mode == 1
        if mode > 1:
            d = y - self.mu[0:3]
            sigma_rho = self.Sigma[pos+5, pos+5]      # a variance where a std-dev is meant, vR.cpp:717
            t = T(np.dot(d, m))
            L_d = T(4) * sigma_rho * abs(t) / (ro*ro*T(np.dot(d, d)))
            want = bool(L_d < T(0.01))
        if not want:
            return y, None, False
        J = np.zeros((3, 6), dtype=T)
        J[:, 0:3] = np.eye(3, dtype=T)
        J[:, 3] = [np.cos(theta)*np.cos(phi)/ro, 0, -np.sin(theta)*np.cos(phi)/ro]
        J[:, 4] = [-np.sin(theta)*np.sin(phi)/ro, -np.cos(phi)/ro, -np.cos(theta)*np.sin(phi)/ro]
        J[:, 5] = -m / (ro*ro)
        return y, J, True

    def convert2xyz_if_linear(self, index) -> bool:
        T = self.T
        ft = self.features[index]
        if ft.coding == XYZ:
            return False
        pos = ft.position_in_state
        y, J_y, conv = self.inverse_depth_to_xyz_world(self.mu[pos:pos+6], 2, pos)
        if not conv:
            return False
        self._convert_covariance(pos, J_y)
        self.mu = np.concatenate([self.mu[:pos], y, self.mu[pos+6:]])
        ft.coding = XYZ
        for g in self.features[index+1:]:
            g.position_in_state -= 3
        return True

    def _convert_covariance(self, pos, J_y):
        T = self.T
        n = self.n
        J = np.zeros((n-3, n), dtype=T)
        J[:pos, :pos] = np.eye(pos, dtype=T)
        J[pos:pos+3, pos:pos+6] = J_y
        J[pos+3:, pos+6:] = np.eye(n-pos-6, dtype=T)
        self.Sigma = J @ self.Sigma @ J.T                                  # vR.cpp:765

    def convert2xyz_if_linear_all(self) -> int:
        c = 0
        for i in range(len(self.features)):                               # vR.cpp:776-780
            if self.features[i].coding == INV and self.convert2xyz_if_linear(i):
                c += 1
        return c

    def feature_xyz(self, index):
        """World point and 3x3 covariance of one feature: inverseDepth2XyzWorld mode 1 and
        Jf Sigma_ff Jf^T as the map export does (RosVSLAMRansac.cpp:177-183)."""
        ft = self.features[index]
        pos = ft.position_in_state
        if ft.coding == XYZ:
            return self.mu[pos:pos+3].copy(), self.Sigma[pos:pos+3, pos:pos+3].copy()
        y, J, _ = self.inverse_depth_to_xyz_world(self.mu[pos:pos+6], 1)
        return y, J @ self.Sigma[pos:pos+6, pos:pos+6] @ J.T


def get_points_features(filt):
    """RosVSLAM::getPointsFeatures (RosVSLAMRansac.cpp:340-418), the table behind points.txt: rows indexed by
    Patch::real_index, (real_index of the LAST live patch) + 1 of them (:350-352); live XYZ features carry
    [X Y Z] * map_scale and their 3x3 covariance block row by row (:376-388), live inverse-depth rows stay zero
    (:363-375); then the patches archived at removal are written over their rows (:406-414), and the archive is
    cleared once it holds more than 7000 entries (:396-404).  An archived patch whose real_index lies beyond the
    table (the reference would write out of bounds there) is skipped."""
    T = filt.T
    if not filt.features:
        return np.zeros((0, 12), dtype=T)
    scale = filt.mu[13] if filt.camera_dim == 14 else T(1)
    rows = filt.features[-1].real_index + 1
    pts = np.zeros((rows, 12), dtype=T)
    for ft in filt.features:
        if ft.coding != XYZ:
            continue
        p = ft.position_in_state
        pts[ft.real_index, 0:3] = filt.mu[p:p+3] * scale
        pts[ft.real_index, 3:12] = filt.Sigma[p:p+3, p:p+3].reshape(-1)
    for (ri, xyz, cov) in filt.deleted_patches:
        if ri < rows:
            pts[ri, 3:12] = cov
            pts[ri, 0:3] = xyz * scale
    if len(filt.deleted_patches) > 7000:
        filt.deleted_patches.clear()
    return pts


def ellipse_parameters(St, sigma_size):
    """computeEllipsoidParameters, vR.cpp:1368-1382, for one 2x2 block: (a, b, theta_deg) ints.
    SelfAdjointEigenSolver returns ascending eigenvalues; the sign of its eigenvectors is not
    specified, the x >= 0 representative is taken here (same ellipse)."""
    St = np.asarray(St, dtype=np.float64)
    w, v = np.linalg.eigh(0.5 * (St + St.T))
    vx, vy = v[0, 0], v[1, 0]
    if vx < 0 or (vx == 0 and vy < 0):
        vx, vy = -vx, -vy
    a = int(sigma_size * math.sqrt(w[0])) if w[0] > 0 else 1
    b = int(sigma_size * math.sqrt(w[1])) if w[1] > 0 else 1
    return a, b, int(180 / 3.14 * math.atan2(vy, vx))


def ransac_1point(filt, z, indices, threshold=None):
    """The hypothesis loop of vR.cpp:986-1034 for EVERY listed feature as the hypothesis (the reference
    draws them at random until its adaptive count runs out).  Returns (counts[M], mask[M(hyp), M(feature)])."""
    T = filt.T
    indices = list(indices)
    M = len(indices)
    z = np.asarray(z, T).reshape(M, 2)
    thr = T(2 * filt.cfg.sigma_pixel if threshold is None else threshold)
    mask = np.zeros((M, M), bool)
    for k, fk in enumerate(indices):
        ft = filt.features[fk]
        H = filt.dense_H([fk])
        S_i = H @ filt.Sigma @ H.T + T(filt.sigma_pixel_2) * np.eye(2, dtype=T)          # :994
        K_i = filt.Sigma @ H.T @ np.linalg.inv(S_i)                                       # :995
        mu_i = filt.mu + K_i @ (z[k] - ft.h)                                               # :996
        r = mu_i[0:3]
        q = mu_i[3:7] / T(np.sqrt(np.dot(mu_i[3:7], mu_i[3:7])))                            # :999
        RotCW = quat2rot(quat_complement(q, T), T)
        for j, fj in enumerate(indices):
            g = filt.features[fj]
            p = g.position_in_state
            if g.coding == INV:
                d, _ = inverse2xyz_projecting(mu_i[p:p+6], r, T, False)
            else:
                d = filt.mu[p:p+3] - r                                                      # mu, not mu_i (:1016)
            hi, _ = filt.cam.project(RotCW @ d, False)
            e = z[j] - hi
            mask[k, j] = bool(T(np.sqrt(np.dot(e, e))) <= thr)                              # :1022
    return mask.sum(axis=1).astype(np.int32), mask


def rescue_high_innovation(filt, mu_before, z, indices, threshold=1.0, return_chi2=False):
    """vR.cpp:1066-1117: h / H of the listed features from the features of the updated state `filt.mu`
    and the camera pose of `mu_before`; S_hi = H Sigma H^T (no R); chi-square gate.  Stores the new
    h / Hc / Hf in the features (patches[i].h / .H, :1089-1110) and returns the is-in-Hi flags."""
    T = filt.T
    z = np.asarray(z, T).reshape(-1, 2)
    r = np.asarray(mu_before[0:3], T)
    q = np.asarray(mu_before[3:7], T)
    out = np.zeros(len(indices), bool)
    chi2 = np.zeros(len(indices), T)
    for k, i in enumerate(indices):
        ft = filt.features[i]
        hi, Hc, Hf, vis, rem = filt.measure_feature(ft, mu=filt.mu, r=r, q=q)
        ft.h, ft.Hc, ft.Hf = hi, Hc, Hf
        H = filt.dense_H([i])
        S_hi = H @ filt.Sigma @ H.T                                                  # :1113
        e = hi - z[k]
        chi2[k] = T(e @ np.linalg.inv(S_hi) @ e)
        out[k] = bool(chi2[k] <= T(threshold))                                       # :1114
    return (out, chi2) if return_chi2 else out


class GlibcRand:
    """glibc's srand(seed) / rand() (random_r TYPE_3: r[i] = r[i-3] + r[i-31] mod 2^32 over a table seeded by the
    Lehmer generator 16807 x mod (2^31 - 1); the first 310 outputs are discarded; rand() = r >> 1).  The reference
    draws its RANSAC hypotheses with srand(time(NULL)) / rand() (vR.cpp:970, 989)."""

    def __init__(self, seed):
        seed = int(seed) & 0xffffffff or 1
        r = [seed]
        for i in range(1, 31):
            prev = r[i - 1] - (1 << 32) if r[i - 1] >= (1 << 31) else r[i - 1]      # as int32
            v = int(np.fmod(16807 * prev, 2147483647))                              # C remainder: sign of the dividend
            if v < 0:
                v += 2147483647
            r.append(v)
        for i in range(31, 34):
            r.append(r[i - 31])
        for i in range(34, 344):
            r.append((r[i - 31] + r[i - 3]) & 0xffffffff)
        self.r = r

    def rand(self):
        r = self.r
        v = (r[-31] + r[-3]) & 0xffffffff
        r.append(v)
        return v >> 1


def update_two_stage(filt, z, indices, plane=False, seed=0, threshold=None, chi2_threshold=1.0):
    """The RANSAC branch of VSlamFilter::update (vR.cpp:964-1130 + 1245-1284) on the matched features `indices`.
    seed = 0: low-innovation set of the best hypothesis (lowest index on ties); seed != 0: the reference's loop --
    draws without replacement from srand(seed) / rand(), adaptive nhyp (:1030), flags of the LAST draw (:1022).
    Returns (is_li, is_hi, hypotheses drawn)."""
    T = filt.T
    indices = list(indices)
    M = len(indices)
    z = np.asarray(z, T).reshape(M, 2)
    li = np.zeros(M, bool)
    hi = np.zeros(M, bool)
    drawn = 0
    if M:
        counts, mask = ransac_1point(filt, z, indices, threshold)
        sel = int(np.argmax(counts))
        drawn = M
        if seed:
            rng = GlibcRand(seed)
            lst = list(range(M))
            nhyp, num_zli, i = 10000, 0, 0
            p = np.float32(0.99)
            drawn = 0
            while i < nhyp and lst:
                sel = lst.pop(rng.rand() % len(lst))                                   # :989-991
                drawn += 1
                if counts[sel] > num_zli:
                    num_zli = int(counts[sel])
                    with np.errstate(divide="ignore"):
                        ratio = np.float32(num_zli) / np.float32(M)
                        den = np.log(np.float32(1) - ratio)
                        val = np.log(np.float32(1) - p) / den                          # :1030
                    nhyp = int(val) if np.isfinite(val) else 0
                i += 1
        li = mask[sel].copy()
    mu_before = filt.mu.copy()
    if li.any():
        sel_i = [indices[k] for k in range(M) if li[k]]
        filt.update(z[li].reshape(-1), sel_i, plane=False)                             # :1036-1064
    rest_k = [k for k in range(M) if not li[k]]
    if rest_k:
        flags = rescue_high_innovation(filt, mu_before, z[rest_k], [indices[k] for k in rest_k], chi2_threshold)
        for t, k in enumerate(rest_k):
            hi[k] = bool(flags[t])
    if hi.any() or plane:
        sel_i = [indices[k] for k in range(M) if hi[k]]
        filt.update(z[hi].reshape(-1), sel_i, plane=plane)                             # :1245-1284
    for k in range(M):                                                                 # update_quality_index, :1297, Patch.cpp:145
        if li[k] or hi[k]:
            filt.features[indices[k]].n_find += 1
    return li, hi, drawn


# --------------------------------------------------------------------------
# structured filter: identical arithmetic, identity blocks exploited
# --------------------------------------------------------------------------
class StructuredFilter(DenseFilter):
    structured = True

    def add_feature(self, u, v) -> int:
        T = self.T
        parts = self._add_feature_parts(u, v)
        if parts is None:
            return 0
        f, G, Jp = parts
        n = self.n
        self.features.append(self._new_feature(n))
        self.mu = np.concatenate([self.mu, f])
        S = np.empty((n+6, n+6), dtype=T)
        S[:n, :n] = self.Sigma
        B = G @ self.Sigma[0:7, :]                     # new rows x old cols
        S[n:, :n] = B
        S[:n, n:] = (self.Sigma[:, 0:7] @ G.T)
        C = G @ self.Sigma[0:7, 0:7] @ G.T + T(self.sigma_pixel_2) * (Jp @ Jp.T)
        C[5, 5] += T(np.float32(self.cfg.sigma_rho_0))
        S[n:, n:] = C
        self.Sigma = S
        return 1

    def add_features(self, pixels) -> int:
        """`add_feature` for a list of pixels, in order, on ONE preallocated covariance (the sequential form copies the
        growing matrix once per feature: 1000 adds at n = 6014 move ~100 GB).  Same expressions on the same operands as
        `add_feature` -- rows n.. = G Sigma[0:7, :n], corner = G Sigma_cc G^T + sigma_px^2 Jp Jp^T (+ sigma_rho_0) --
        so the result is bit-identical to the sequential adds (tests/test_oracle_flavours.py).  Returns the number
        of features added (pixels outside the image are skipped, as `add_feature` returning 0)."""
        T = self.T
        pixels = [tuple(p) for p in pixels]
        n0 = self.n
        cap = n0 + 6 * len(pixels)
        S = np.zeros((cap, cap), dtype=T)
        S[:n0, :n0] = self.Sigma
        mu = np.zeros(cap, dtype=T)
        mu[:n0] = self.mu
        n = n0
        added = 0
        for (u, v) in pixels:
            parts = self._add_feature_parts(u, v)        # reads mu[0:7] only: unchanged by the adds
            if parts is None:
                continue
            f, G, Jp = parts
            self.features.append(self._new_feature(n))
            mu[n:n+6] = f
            S[n:n+6, :n] = G @ S[0:7, :n]
            S[:n, n:n+6] = S[:n, 0:7] @ G.T
            C = G @ S[0:7, 0:7] @ G.T + T(self.sigma_pixel_2) * (Jp @ Jp.T)
            C[5, 5] += T(np.float32(self.cfg.sigma_rho_0))
            S[n:n+6, n:n+6] = C
            n += 6
            added += 1
        self.mu = mu[:n].copy()
        self.Sigma = S[:n, :n].copy()
        return added

    def predict_covariance(self, Ft, Q):
        S = self.Sigma
        S[0:13, :] = Ft @ S[0:13, :]
        S[:, 0:13] = S[:, 0:13] @ Ft.T
        S[0:13, 0:13] += Q

    def _apply_quat_block(self, Q):
        S = self.Sigma
        S[3:7, :] = Q @ S[3:7, :]
        S[:, 3:7] = S[:, 3:7] @ Q.T

    # The three per-feature loops below (measure, Sigma H^T, H X) are batched over the features: same formulas,
    # same element type, one numpy call per term instead of one Python iteration per feature (the N = 200,
    # 1000-frame parity test would otherwise spend minutes in the interpreter).  DenseFilter keeps the
    # line-by-line loops of the reference; tests/test_oracle_flavours.py holds the two flavours together.
    def _compact(self, indices):
        """Hc (M,2,7), Hf zero-padded to (M,2,6), state columns (M,6) of the listed features (padding columns of an
        XYZ feature point at a valid index and meet a zero of Hf)."""
        T = self.T
        M = len(indices)
        Hc = np.empty((M, 2, 7), dtype=T)
        Hf = np.zeros((M, 2, 6), dtype=T)
        cols = np.empty((M, 6), dtype=np.int64)
        for k, i in enumerate(indices):
            ft = self.features[i]
            Hc[k] = ft.Hc
            Hf[k, :, :ft.size] = ft.Hf
            cols[k] = ft.position_in_state + np.minimum(np.arange(6), ft.size - 1)
        return Hc, Hf, cols

    def sigma_Ht(self, indices, plane=False):
        """W = Sigma H^T (n x m) from compact H: column pair k = Sigma[:,0:7] Hc_k^T + Sigma[:,p_k:p_k+size] Hf_k^T."""
        T = self.T
        M = len(indices)
        m = 2*M + (3 if plane else 0)
        W = np.empty((self.n, m), dtype=T)
        S = self.Sigma
        if M:
            Hc, Hf, cols = self._compact(indices)
            Wc = S[:, 0:7] @ Hc.reshape(2*M, 7).T
            Wf = np.einsum('nkt,kat->nka', S[:, cols], Hf).reshape(self.n, 2*M)
            W[:, :2*M] = Wc + Wf
        if plane:
            W[:, m-3] = S[:, 1]
            W[:, m-2] = S[:, 4]
            W[:, m-1] = S[:, 6]
        return W

    def H_times(self, X, indices, plane=False):
        """H X for X with n rows, from compact H: row pair k = Hc_k X[0:7] + Hf_k X[p_k:p_k+size]."""
        T = self.T
        M = len(indices)
        m = 2*M + (3 if plane else 0)
        out = np.empty((m, X.shape[1]), dtype=T)
        if M:
            Hc, Hf, cols = self._compact(indices)
            oc = Hc.reshape(2*M, 7) @ X[0:7, :]
            of = np.einsum('kat,ktc->kac', Hf, X[cols, :]).reshape(2*M, X.shape[1])
            out[:2*M, :] = oc + of
        if plane:
            out[m-3, :] = X[1, :]
            out[m-2, :] = X[4, :]
            out[m-1, :] = X[6, :]
        return out

    def measure(self):
        """The loop of vR.cpp:508-579 for every feature at once (see DenseFilter.measure_feature for the
        per-feature restatement with the reference's line numbers); then the position_in_z pass (:584-592)."""
        T = self.T
        cfg = self.cfg
        cam = self.cam
        N = len(self.features)
        if N:
            mu = self.mu
            r = mu[0:3]
            qc = quat_complement(mu[3:7], T)
            Rcw = quat2rot(qc, T)
            pos = np.array([ft.position_in_state for ft in self.features])
            inv = np.array([ft.coding == INV for ft in self.features])
            F = mu[pos[:, None] + np.where(inv[:, None], np.arange(6), np.minimum(np.arange(6), 2))]   # (N,6)
            theta, phi, ro = F[:, 3], F[:, 4], F[:, 5]
            st, ct, sp_, cp = np.sin(theta), np.cos(theta), np.sin(phi), np.cos(phi)
            m = np.stack([st*cp, -sp_, ct*cp], axis=1).astype(T)
            ar = F[:, 0:3] - r
            d = np.where(inv[:, None], ro[:, None] * ar + m, ar).astype(T)
            Jf = np.zeros((N, 3, 6), dtype=T)                    # d(d)/d(feature): inverse2XYZ4_projecting's J / I3
            eye = np.eye(3, dtype=T)
            Jf[:, :, 0:3] = np.where(inv[:, None, None], ro[:, None, None] * eye, eye)
            z0 = np.zeros(N, dtype=T)
            Jf[:, :, 3] = np.where(inv[:, None], np.stack([ct*cp, z0, -st*cp], axis=1), 0)
            Jf[:, :, 4] = np.where(inv[:, None], np.stack([-st*sp_, -cp, -ct*sp_], axis=1), 0)
            Jf[:, :, 5] = np.where(inv[:, None], ar, 0)
            scale_r = np.where(inv, -ro, T(-1)).astype(T)
            rem = inv & (ro <= 0)
            hC = (d @ Rcw.T).astype(T)
            x, y, z = hC[:, 0], hC[:, 1], hC[:, 2]
            x1, y1 = x / z, y / z
            r2 = x1*x1 + y1*y1
            L = T(1) + cam.k1*r2 + cam.k2*r2*r2 + cam.k3*r2*r2*r2
            x2 = x1*L + T(2)*cam.p1*x1*y1 + cam.p2*(r2 + T(2)*x1*x1)
            y2 = y1*L + T(2)*cam.p2*x1*y1 + cam.p1*(r2 + T(2)*y1*y1)
            h = np.stack([cam.fx*x2 + cam.u0, cam.fy*y2 + cam.v0], axis=1).astype(T)
            fpoly = cam.k1 + T(2)*cam.k2*r2 + T(3)*cam.k3*r2*r2
            D = np.empty((N, 2, 2), dtype=T)                     # diff_distort_undistort, cam.cpp:18-47
            D[:, 0, 0] = L + T(2)*fpoly*x1*x1 + T(2)*cam.p1*y1 + T(2)*cam.p2*x1 + T(4)*cam.p2*x1
            D[:, 0, 1] = T(2)*fpoly*x1*y1 + T(2)*cam.p1*x1 + T(2)*cam.p2*y1
            D[:, 1, 0] = T(2)*fpoly*y1*x1 + T(2)*cam.p2*y1 + T(2)*cam.p1*x1
            D[:, 1, 1] = L + T(2)*fpoly*y1*y1 + T(2)*cam.p2*x1 + T(2)*cam.p1*y1 + T(4)*cam.p1*y1
            Jn = np.zeros((N, 2, 3), dtype=T)
            Jn[:, 0, 0] = T(1)/z
            Jn[:, 0, 2] = -x/z/z
            Jn[:, 1, 1] = T(1)/z
            Jn[:, 1, 2] = -y/z/z
            Jp = np.array([cam.fx, cam.fy], dtype=T)[None, :, None] * (D @ Jn)       # J_h_hC, (N,2,3)
            half = int(cfg.window_size) // 2
            inside = (h[:, 0] > half) & (h[:, 1] > half) & (h[:, 0] < cfg.image_width - half) & \
                     (h[:, 1] < cfg.image_height - half)
            vis = (~rem) & inside & (z >= 0)
            dR = np.stack([diff_quat2rot(qc, j, T) for j in range(4)])               # (4,3,3)
            J_hC_q = np.einsum('jab,nb->naj', dR, d) * np.array([1, -1, -1, -1], dtype=T)   # (N,3,4) incl. d_qbar_q
            JR = Jp @ Rcw                                                            # (N,2,3)
            Hc = np.empty((N, 2, 7), dtype=T)
            Hc[:, :, 0:3] = scale_r[:, None, None] * JR
            Hc[:, :, 3:7] = Jp @ J_hC_q
            Hf = JR @ Jf                                                             # (N,2,6)
            for i, ft in enumerate(self.features):
                if rem[i]:
                    ft.remove_flag = True
                ft.is_in_innovation = bool(vis[i])
                ft.h, ft.Hc, ft.Hf = h[i].copy(), Hc[i].copy(), Hf[i, :, :ft.size].copy()
        j = 0
        for ft in self.features:                                        # vR.cpp:584-592
            if ft.is_in_innovation:
                ft.position_in_z = 2 * j
                j += 1
        vis_idx = self.visible_indices()
        self.h_out = (np.concatenate([self.features[i].h for i in vis_idx]) if vis_idx
                      else np.zeros(0, dtype=self.T))

    def innovation_covariance(self, indices, plane=False):
        T = self.T
        if not indices and not plane:
            return np.zeros((0, 0), dtype=T)
        W = self.sigma_Ht(indices, plane)
        St = self.H_times(W, indices, plane)
        p = St.shape[0]
        R = T(self.sigma_pixel_2) * np.ones(p, dtype=T)
        if plane:
            R[p-3:] = T(0.00001)
        St[np.arange(p), np.arange(p)] += R
        return St

    def _update_block(self, indices, plane, z, h):
        T = self.T
        W = self.sigma_Ht(indices, plane)
        St = self.H_times(W, indices, plane)
        p = St.shape[0]
        R = T(self.sigma_pixel_2) * np.ones(p, dtype=T)
        if plane:
            R[p-3:] = T(0.00001)
        St[np.arange(p), np.arange(p)] += R
        Kt = W @ np.linalg.inv(St)
        self.mu = self.mu + Kt @ (z - h)
        HS = self.H_times(self.Sigma, indices, plane)     # H Sigma (not W^T: Sigma is not kept symmetric)
        self.Sigma = self.Sigma - Kt @ HS
        self.St, self.Kt = St, Kt

    def _convert_covariance(self, pos, J_y):
        S = self.Sigma
        n = self.n
        keep_r = np.r_[0:pos+3, pos+6:n]
        A = S.copy()
        A[pos:pos+3, :] = J_y @ S[pos:pos+6, :]
        A = A[keep_r, :]
        B = A.copy()
        B[:, pos:pos+3] = A[:, pos:pos+6] @ J_y.T
        self.Sigma = B[:, keep_r]


# --------------------------------------------------------------------------
# synthetic scenario: seeds 1234 / 1235 / 1236
# --------------------------------------------------------------------------
def synthetic_pixels(cfg: Config, n_features: int, seed: int = 1234):
    """N pixels uniform inside the add-feature window margin."""
    rng = np.random.default_rng(seed)
    half = cfg.window_size // 2
    u = rng.uniform(half + 1, cfg.image_width - half - 1, size=n_features)
    v = rng.uniform(half + 1, cfg.image_height - half - 1, size=n_features)
    return np.stack([u, v], axis=1)


def synthetic_measurements(filt: DenseFilter, indices, seed: int = 1235, sigma=None):
    """z = h(mu) + N(0, sigma_px^2) for the listed features (after predict())."""
    rng = np.random.default_rng(seed)
    sigma = float(filt.cfg.sigma_pixel) if sigma is None else sigma
    h = filt.stacked_h(list(indices)).astype(np.float64)
    return (h + rng.normal(0.0, sigma, size=h.shape)).astype(filt.T)


def build_scenario(flavour, cfg: Config, n_features: int, dtype=np.float32, seed=1234,
                   v=(0.3, 0.0, 0.0), w=(0.0, 0.05, 0.0), dT=1.0/30.0, camera_dim=STATE_DIM, batched_add=False):
    """Initial state of SURVEY 8d: mu0/Sigma0 of the constructor, constant velocity set in mu,
    N features inserted through the add-feature math in order."""
    filt = flavour(cfg, dtype, camera_dim)
    filt.dT = dT
    filt.mu[7:10] = np.asarray(v, dtype=filt.T)
    filt.mu[10:13] = np.asarray(w, dtype=filt.T)
    px = synthetic_pixels(cfg, n_features, seed)
    if batched_add and hasattr(filt, "add_features"):
        assert filt.add_features(px) == n_features
        return filt
    for (pu, pv) in px:
        ok = filt.add_feature(pu, pv)
        assert ok == 1
    return filt
