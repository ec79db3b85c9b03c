"""`VSlamFilter`: host-side mirror of the reference filter class over the C ABI.

Method names and argument meaning follow mono-slam/src/vslamRansac.hpp:94-141 so parity
tests read like calls on the reference class.  Image-side methods (captureNewFrame with a
cv::Mat, findNewFeatures, drawing) are out of scope; `captureNewFrame` only keeps the
time-stamp logic (vR.cpp:226-233).  Eigen matrices become numpy arrays (column-major data
from the ABI is returned as ordinary (row, col) arrays).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import capi
from .capi import EkfConfig, EkfError

STATE_DIM = 14


def _cfg_from(**kw) -> EkfConfig:
    lib = capi.load_library()
    c = EkfConfig()
    lib.ekf_config_default(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def kinect_config() -> dict:
    """mono-slam/conf/conf_kinect.cfg with scale = 2 applied (ConfigVSLAM.cpp:87-103)."""
    s = 2.0
    return dict(sigma_vx=0.03, sigma_vy=0.03, sigma_vz=0.03, sigma_wx=0.015, sigma_wy=0.015,
                sigma_wz=0.015, rho_0=0.2, sigma_rho_0=0.25, sigma_pixel=2, window_size=15,
                kernel_size=1000, scale=2, T_camera=0.0, nInitFeatures=50, min_features=100,
                max_features=1000, fx=537.673722507338 / s, fy=534.380205679756 / s,
                u0=321.226061527052 / s, v0=249.773992466202 / s, k1=0.0395956005042652,
                k2=-0.111310452999064, k3=0.0, p1=0.00211988964071199, p2=0.00070924348636878,
                image_width=320, image_height=240)


def sim_config() -> dict:
    """conf_sim.cfg with scale = 10 applied."""
    s = 10.0
    return dict(sigma_vx=0.0000008, sigma_vy=0.0000008, sigma_vz=0.00000000008,
                sigma_wx=0.000000004, sigma_wy=0.0000004, sigma_wz=0.00000000004, rho_0=0.1,
                sigma_rho_0=0.25, sigma_pixel=2, window_size=30, kernel_size=3, scale=10,
                T_camera=0.2, sigma_size=4, min_features=20, max_features=35,
                fx=2217.0187 / s, fy=2217.0187 / s, u0=1280.5 / s, v0=960.5 / s,
                k1=0.0, k2=0.0, k3=0.0, p1=0.0, p2=0.0, image_width=256, image_height=192)


class VSlamFilter:
    """Drop-in shaped like the reference `VSlamFilter` (math methods only)."""

    def __init__(self, config: Optional[dict] = None, capacity_features: int = 1024,
                 dtype=np.float32, camera_dim: int = STATE_DIM, device: int = 0):
        self._lib = capi.load_library()
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise EkfError(1, "dtype must be float32 or float64")
        self._cfg = _cfg_from(**(config or {}))
        self._h = C.c_void_p()
        rc = self._lib.ekf_create(C.byref(self._cfg), camera_dim, capacity_features,
                                  capi.EKF_F32 if self.dtype == np.float32 else capi.EKF_F64,
                                  device, C.byref(self._h))
        if rc != 0:
            msg = self._lib.ekf_last_error(None)
            raise EkfError(rc, msg.decode() if msg else "ekf_create failed")
        self.camera_dim = camera_dim
        self._old_ts = -1.0

    # -- plumbing ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.ekf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.ekf_last_error(self._h)
            raise EkfError(rc, msg.decode() if msg else "")

    def _count(self, rc):
        if rc < 0:
            msg = self._lib.ekf_last_error(self._h)
            raise EkfError(-rc, msg.decode() if msg else "")
        return rc

    @staticmethod
    def _ptr(a):
        return a.ctypes.data_as(C.c_void_p)

    def set_option(self, option: int, value: int):
        self._check(self._lib.ekf_set_option(self._h, option, value))

    def set_stream(self, hip_stream: int):
        self._check(self._lib.ekf_set_stream(self._h, C.c_void_p(hip_stream)))

    def synchronize(self):
        self._check(self._lib.ekf_synchronize(self._h))

    # -- reference API ------------------------------------------------------------------------
    def captureNewFrame(self, time_stamp: Optional[float] = None):
        """Time-stamp half of captureNewFrame (vR.cpp:226-233): sets dT."""
        if time_stamp is not None:
            if self._old_ts > 0:
                self.setDt(time_stamp - self._old_ts)
            self._old_ts = time_stamp

    def setDt(self, dT: float):
        self._check(self._lib.ekf_set_dt(self._h, float(dT)))

    def getDt(self) -> float:
        return self._lib.ekf_get_dt(self._h)

    def addFeature(self, pf: Sequence[float]) -> int:
        return self._count(self._lib.ekf_add_feature(self._h, float(pf[0]), float(pf[1])))

    def removeFeature(self, index: int):
        self._check(self._lib.ekf_remove_feature(self._h, int(index)))

    def removeFeatures(self, indices: Sequence[int]):
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        self._check(self._lib.ekf_remove_features(self._h, self._ptr(idx), idx.size))

    def predict(self, Translation_Speed_Control=None, Rotational_Speed_Control=None, Vcontrol=False):
        t = None if Translation_Speed_Control is None else np.ascontiguousarray(Translation_Speed_Control, self.dtype)
        r = None if Rotational_Speed_Control is None else np.ascontiguousarray(Rotational_Speed_Control, self.dtype)
        self._check(self._lib.ekf_predict(self._h, None if t is None else self._ptr(t),
                                          None if r is None else self._ptr(r), int(bool(Vcontrol))))

    def measure(self):
        self._check(self._lib.ekf_measure(self._h))

    def motionJacobian(self):
        """Ft (13,13) of System_model_jacobian (vR.cpp:1492-1507) and Q (13,13) (vR.cpp:463-475) of the last predict."""
        Ft = np.zeros((13, 13), self.dtype)
        Q = np.zeros((13, 13), self.dtype)
        self._check(self._lib.ekf_get_motion_jacobian(self._h, self._ptr(Ft), self._ptr(Q)))
        return Ft.T.copy(), Q.T.copy()

    def predictions(self, jacobians: bool = False):
        """h (N,2), visible (N,), remove (N,), S2x2 (N,2,2) [, Hc (N,2,7), Hf (N,2,6)]."""
        N = self.numOfFeatures()
        h = np.zeros((N, 2), self.dtype)
        vis = np.zeros(N, np.uint8)
        rem = np.zeros(N, np.uint8)
        s2 = np.zeros((N, 4), self.dtype)
        hc = np.zeros((N, 14), self.dtype) if jacobians else None
        hf = np.zeros((N, 12), self.dtype) if jacobians else None
        self._check(self._lib.ekf_get_predictions(
            self._h, self._ptr(h), self._ptr(vis), self._ptr(rem), self._ptr(s2),
            None if hc is None else self._ptr(hc), None if hf is None else self._ptr(hf)))
        S = s2.reshape(N, 2, 2).transpose(0, 2, 1)          # column-major 2x2 -> (row, col)
        out = [h, vis.astype(bool), rem.astype(bool), S]
        if jacobians:
            out += [hc.reshape(N, 7, 2).transpose(0, 2, 1), hf.reshape(N, 6, 2).transpose(0, 2, 1)]
        return tuple(out)

    def update(self, z=None, indices=None, plane_constraint: Optional[bool] = None):
        """The EKF update block (vR.cpp:1245-1284) on the measured set `indices` with pixels z."""
        plane = bool(self._cfg.forsePlane) if plane_constraint is None else bool(plane_constraint)
        if z is None:
            z = np.zeros(0, self.dtype)
            indices = []
        z = np.ascontiguousarray(z, self.dtype).reshape(-1)
        idx = np.ascontiguousarray(indices, np.int32)
        if z.size != 2 * idx.size:
            raise EkfError(1, "z must hold 2 values per listed feature")
        self._check(self._lib.ekf_update(self._h, self._ptr(z), self._ptr(idx), idx.size, int(plane)))

    def update_device(self, d_z: int, d_indices: int, M: int, plane_constraint: bool = False):
        self._check(self._lib.ekf_update_device(self._h, C.c_void_p(d_z), C.c_void_p(d_indices), int(M),
                                                int(bool(plane_constraint))))

    def innovationCovariance(self, indices, plane_constraint: bool = False):
        idx = np.ascontiguousarray(indices, np.int32)
        m = 2 * idx.size + (3 if plane_constraint else 0)
        out = np.zeros((m, m), self.dtype)
        self._check(self._lib.ekf_innovation_covariance(self._h, self._ptr(idx), idx.size,
                                                        int(bool(plane_constraint)), self._ptr(out)))
        return out.T.copy()

    def rescueHighInnovation(self, cam_before, z, indices, chi2_threshold: float = 1.0):
        """High-innovation rescue (vR.cpp:1066-1117): re-linearise the listed features and gate them."""
        cam = np.ascontiguousarray(cam_before, self.dtype).reshape(-1)[:7].copy()
        z = np.ascontiguousarray(z, self.dtype).reshape(-1)
        idx = np.ascontiguousarray(indices, np.int32)
        out = np.zeros(idx.size, np.uint8)
        if idx.size:
            self._check(self._lib.ekf_rescue_high_innovation(self._h, self._ptr(cam), self._ptr(z), self._ptr(idx),
                                                             idx.size, float(chi2_threshold), self._ptr(out)))
        return out.astype(bool)

    # ---- image side (Patch.cpp / libblur.cpp) ---------------------------------------------------
    def setFrame(self, gray):
        """captureNewFrame's image after resize / grayscale: uint8 (height, width)."""
        g = np.ascontiguousarray(gray, np.uint8)
        if g.ndim != 2:
            raise ValueError("frame must be a single-channel 8-bit image")
        self._check(self._lib.ekf_set_frame(self._h, self._ptr(g), g.shape[1], g.shape[0], g.strides[0]))

    def setPatch(self, index: int, pixels):
        p = np.ascontiguousarray(pixels, np.uint8).reshape(-1)
        self._check(self._lib.ekf_set_patch(self._h, int(index), self._ptr(p)))

    def getPatch(self, index: int, matching: bool = False):
        w = int(self._cfg.window_size)
        out = np.zeros((w, w), np.uint8)
        self._check(self._lib.ekf_get_patch(self._h, int(index), 1 if matching else 0, self._ptr(out)))
        return out

    def blurPredictions(self):
        out = np.zeros((self.numOfFeatures(), 2), self.dtype)
        self._check(self._lib.ekf_get_blur_predictions(self._h, self._ptr(out)))
        return out

    def findMatches(self, threshold: float = 0.8):
        """Patch::findMatch for every visible feature: (z (N,2), found (N,) bool, score (N,) float32)."""
        N = self.numOfFeatures()
        z = np.zeros((N, 2), self.dtype)
        found = np.zeros(N, np.uint8)
        score = np.zeros(N, np.float32)
        self._check(self._lib.ekf_find_matches(self._h, float(threshold), self._ptr(z), self._ptr(found), self._ptr(score)))
        return z, found.astype(bool), score

    def getPointsFeatures(self, convert_inverse_depth: bool = False):
        """RosVSLAM::getPointsFeatures (RosVSLAMRansac.cpp:340-418): (N, 12) = xyz * map_scale + 3x3 covariance."""
        out = np.zeros((self.numOfFeatures(), 12), self.dtype)
        self._check(self._lib.ekf_export_points(self._h, self._ptr(out), int(bool(convert_inverse_depth))))
        return out

    def getPointsTable(self):
        """RosVSLAM::getPointsFeatures in the reference's own layout (RosVSLAMRansac.cpp:340-418): rows indexed by
        Patch::real_index, the patches archived at removal (vR.cpp:394-404) included -- what `points.txt` holds."""
        rows = C.c_int(0)
        self._check(self._lib.ekf_export_points_table(self._h, None, 0, C.byref(rows)))
        out = np.zeros((rows.value, 12), self.dtype)
        if rows.value:
            self._check(self._lib.ekf_export_points_table(self._h, self._ptr(out), rows.value, C.byref(rows)))
        return out

    def featureIds(self):
        """(real_index, n_find) per live feature: Patch::real_index / Patch::n_find."""
        N = self.numOfFeatures()
        ri, nf = np.zeros(N, np.int32), np.zeros(N, np.int32)
        self._check(self._lib.ekf_get_feature_ids(self._h, self._ptr(ri), self._ptr(nf)))
        return ri, nf

    def setFeatureMeta(self, index: int, real_index: int = -1, n_find: int = -1):
        self._check(self._lib.ekf_set_feature_meta(self._h, int(index), int(real_index), int(n_find)))

    def numArchived(self) -> int:
        return int(self._lib.ekf_num_archived(self._h))

    def searchEllipses(self, sigma_size: Optional[int] = None):
        """computeEllipsoidParameters (vR.cpp:1368-1382): (N,3) ints (a, b, theta_deg) per feature."""
        N = self.numOfFeatures()
        out = np.zeros((N, 3), np.int32)
        ss = int(self._cfg.sigma_size if sigma_size is None else sigma_size)
        self._check(self._lib.ekf_get_search_ellipses(self._h, ss, self._ptr(out)))
        return out

    def ransac1Point(self, z, indices, threshold: Optional[float] = None):
        """All 1-point RANSAC hypotheses at once (vR.cpp:986-1034): (counts (M,), best k, inliers of best (M,))."""
        z = np.ascontiguousarray(z, self.dtype).reshape(-1)
        idx = np.ascontiguousarray(indices, np.int32)
        thr = 2.0 * self._cfg.sigma_pixel if threshold is None else float(threshold)
        counts = np.zeros(idx.size, np.int32)
        inl = np.zeros(idx.size, np.uint8)
        best = C.c_int()
        self._check(self._lib.ekf_ransac_1point(self._h, self._ptr(z), self._ptr(idx), idx.size, thr,
                                                self._ptr(counts), self._ptr(inl), C.byref(best)))
        return counts, best.value, inl.astype(bool)

    def updateTwoStage(self, z, indices, plane_constraint: Optional[bool] = None, seed: int = 0,
                       ransac_threshold: Optional[float] = None, chi2_threshold: float = 1.0):
        """The RANSAC branch of update() (vR.cpp:964-1130 + 1245-1284) in one call: (is_li (M,), is_hi (M,), draws)."""
        plane = bool(self._cfg.forsePlane) if plane_constraint is None else bool(plane_constraint)
        z = np.ascontiguousarray(z, self.dtype).reshape(-1)
        idx = np.ascontiguousarray(indices, np.int32)
        thr = 2.0 * self._cfg.sigma_pixel if ransac_threshold is None else float(ransac_threshold)
        li = np.zeros(idx.size, np.uint8)
        hi = np.zeros(idx.size, np.uint8)
        drawn = C.c_int()
        self._check(self._lib.ekf_update_two_stage(self._h, self._ptr(z), self._ptr(idx), idx.size, int(plane),
                                                   int(seed), thr, float(chi2_threshold), self._ptr(li), self._ptr(hi),
                                                   C.byref(drawn)))
        return li.astype(bool), hi.astype(bool), drawn.value

    def getGain(self):
        m = self._lib.ekf_last_measurement_rows(self._h)
        n = self.stateDim()
        out = np.zeros((m, n), self.dtype)                   # column-major n x m
        self._check(self._lib.ekf_get_gain(self._h, self._ptr(out)))
        return out.T.copy()

    def convert2XYZ_ifLinear(self, index: int) -> int:
        return self._count(self._lib.ekf_convert_xyz_if_linear(self._h, int(index)))

    def convert2XYZ_ifLinearAll(self) -> int:
        return self._count(self._lib.ekf_convert_xyz_if_linear_all(self._h))

    def numOfFeatures(self) -> int:
        return self._lib.ekf_num_features(self._h)

    def stateDim(self) -> int:
        return self._lib.ekf_state_dim(self._h)

    def featureLayout(self):
        N = self.numOfFeatures()
        pos = np.zeros(N, np.int32)
        cod = np.zeros(N, np.int32)
        self._check(self._lib.ekf_get_feature_layout(self._h, self._ptr(pos), self._ptr(cod)))
        return pos, cod

    def getState(self):
        """mu[0:STATE_DIM] (vR.cpp:135-140)."""
        return self.getFullState()[:self.camera_dim]

    def getFullState(self):
        n = self.stateDim()
        out = np.zeros(n, self.dtype)
        self._check(self._lib.ekf_get_state(self._h, self._ptr(out), 0, n))
        return out

    def setFullState(self, mu):
        mu = np.ascontiguousarray(mu, self.dtype)
        self._check(self._lib.ekf_set_state(self._h, self._ptr(mu), 0, mu.size))

    def setStateSegment(self, offset: int, values):
        v = np.ascontiguousarray(values, self.dtype).reshape(-1)
        self._check(self._lib.ekf_set_state(self._h, self._ptr(v), int(offset), v.size))

    def getSigma(self):
        """Sigma[0:STATE_DIM, 0:STATE_DIM] (vR.cpp:131-133)."""
        return self.getSigmaBlock(0, 0, self.camera_dim, self.camera_dim)

    def getSigmaBlock(self, r0, c0, rows, cols):
        out = np.zeros((cols, rows), self.dtype)             # column-major rows x cols
        self._check(self._lib.ekf_get_sigma_block(self._h, self._ptr(out), r0, c0, rows, cols))
        return out.T.copy()

    def peekWorkspace(self, which, r0, c0, rows, cols):
        """Diagnostics: rows x cols of W (which = 0) or V (1) as the last update left them (ekf_peek_workspace)."""
        out = np.zeros((rows, cols), self.dtype)
        self._check(self._lib.ekf_peek_workspace(self._h, int(which), self._ptr(out), r0, c0, rows, cols))
        return out

    def getFullSigma(self):
        n = self.stateDim()
        return self.getSigmaBlock(0, 0, n, n)

    def setSigmaBlock(self, block, r0=0, c0=0):
        block = np.asarray(block, self.dtype)
        cm = np.ascontiguousarray(block.T)
        self._check(self._lib.ekf_set_sigma_block(self._h, self._ptr(cm), r0, c0, block.shape[0], block.shape[1]))

    def Covariance_Parameter(self) -> float:
        out = C.c_double()
        self._check(self._lib.ekf_covariance_parameter(self._h, C.byref(out)))
        return out.value

    def checkInvariants(self):
        """(max |Sigma| outside the live block of the padded device buffer, max |Sigma - Sigma^T|, max |Sigma|)."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self._check(self._lib.ekf_check_invariants(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def featureXYZ(self, index: int):
        """inverseDepth2XyzWorld(mode 1) and Jf Sigma Jf^T (RosVSLAMRansac.cpp:177-183)."""
        xyz = np.zeros(3, self.dtype)
        cov = np.zeros((3, 3), self.dtype)
        self._check(self._lib.ekf_feature_xyz(self._h, int(index), self._ptr(xyz), self._ptr(cov)))
        return xyz, cov.T.copy()

    # -- profiling ---------------------------------------------------------------------------
    def profile(self):
        out = {}
        for k in range(self._lib.ekf_profile_kernels()):
            ms = C.c_double()
            cnt = C.c_longlong()
            self._check(self._lib.ekf_profile_read(self._h, k, C.byref(ms), C.byref(cnt)))
            if cnt.value:
                out[self._lib.ekf_profile_kernel_name(k).decode()] = (ms.value, cnt.value)
        return out

    def profile_work(self):
        """Algorithmic flop of the launches timed under each kernel name (kept for "downdate_syrk")."""
        out = {}
        for k in range(self._lib.ekf_profile_kernels()):
            w = C.c_double()
            self._check(self._lib.ekf_profile_work(self._h, k, C.byref(w)))
            if w.value:
                out[self._lib.ekf_profile_kernel_name(k).decode()] = w.value
        return out

    def chunkPlan(self):
        """(block rows, [block step at which each column chunk of the last update ended], W re-evaluated?)."""
        ends = (C.c_int * 8)()
        block, wrec = C.c_int(0), C.c_int(0)
        k = int(self._lib.ekf_get_chunk_plan(self._h, ends, 8, C.byref(block), C.byref(wrec)))
        return block.value, [int(ends[g]) for g in range(min(k, 8))], bool(wrec.value)

    def launch_counts(self):
        """{launch kind: launches since creation / profile_reset} (ekf_launch_count): which launch structure the updates
        of this handle actually took -- e.g. "downdate_bf16x6" against "downdate_f32"."""
        out = {}
        for k in range(self._lib.ekf_launch_kinds()):
            cnt = C.c_longlong()
            self._check(self._lib.ekf_launch_count(self._h, k, C.byref(cnt)))
            out[self._lib.ekf_launch_kind_name(k).decode()] = int(cnt.value)
        return out

    def profile_reset(self):
        self._check(self._lib.ekf_profile_reset(self._h))

    def device_pointers(self):
        ld = C.c_int()
        s = self._lib.ekf_device_sigma(self._h, C.byref(ld))
        return self._lib.ekf_device_mu(self._h), s, ld.value
