"""ctypes binding of include/ekf_monoslam.h (one prototype per declared entry point)."""
from __future__ import annotations

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EKF_LIB_PATH") or os.path.join(HERE, "lib", "libekfslam_hip.so")   # (EKF_LIB_PATH: an A/B build, tools/ only)
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "ekf_monoslam.h")

EKF_F32, EKF_F64 = 0, 1
EKF_OPT_PROPAGATE_STREAMING, EKF_OPT_USE_MFMA, EKF_OPT_PROFILE, EKF_OPT_PIPELINE = 0, 1, 2, 3
EKF_OPT_SPLIT_BF16, EKF_OPT_FEATURE_NOISE, EKF_OPT_FUSED_LAUNCHES, EKF_OPT_W_RECOMPUTE = 4, 5, 6, 7
STATUS_NAMES = {0: "EKF_OK", 1: "EKF_ERR_ARG", 2: "EKF_ERR_CAPACITY", 3: "EKF_ERR_DEVICE",
                4: "EKF_ERR_STATE", 5: "EKF_ERR_NUMERIC", 6: "EKF_ERR_UNSUPPORTED"}


class EkfError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class EkfConfig(C.Structure):
    """struct ekf_config (ConfigVSLAM.h:23-48 + camModel.hpp:9-11 + frame size)."""
    _fields_ = [
        ("sigma_vx", C.c_float), ("sigma_vy", C.c_float), ("sigma_vz", C.c_float),
        ("sigma_wx", C.c_float), ("sigma_wy", C.c_float), ("sigma_wz", C.c_float),
        ("rho_0", C.c_float), ("sigma_rho_0", C.c_float),
        ("window_size", C.c_int), ("sigma_pixel", C.c_int), ("kernel_size", C.c_int),
        ("sigma_size", C.c_int), ("scale", C.c_int),
        ("T_camera", C.c_float),
        ("nInitFeatures", C.c_int), ("min_features", C.c_int), ("max_features", C.c_int),
        ("forsePlane", C.c_int),
        ("fx", C.c_float), ("fy", C.c_float), ("u0", C.c_float), ("v0", C.c_float),
        ("k1", C.c_float), ("k2", C.c_float), ("k3", C.c_float), ("p1", C.c_float), ("p2", C.c_float),
        ("image_width", C.c_int), ("image_height", C.c_int),
    ]


def declared_symbols(header_path: str = HEADER_PATH):
    """Names of every function the header declares (used by the ABI export test)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ekf_[a-z0-9_]+)\s*\(", text)))


_P = C.c_void_p
_PROTOS = {
    "ekf_config_default": (None, [C.POINTER(EkfConfig)]),
    "ekf_abi_version": (C.c_int, []),
    "ekf_create": (C.c_int, [C.POINTER(EkfConfig), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "ekf_destroy": (None, [_P]),
    "ekf_last_error": (C.c_char_p, [_P]),
    "ekf_set_dt": (C.c_int, [_P, C.c_double]),
    "ekf_get_dt": (C.c_double, [_P]),
    "ekf_set_stream": (C.c_int, [_P, _P]),
    "ekf_set_option": (C.c_int, [_P, C.c_int, C.c_int]),
    "ekf_synchronize": (C.c_int, [_P]),
    "ekf_add_feature": (C.c_int, [_P, C.c_double, C.c_double]),
    "ekf_remove_feature": (C.c_int, [_P, C.c_int]),
    "ekf_remove_features": (C.c_int, [_P, _P, C.c_int]),
    "ekf_predict": (C.c_int, [_P, _P, _P, C.c_int]),
    "ekf_measure": (C.c_int, [_P]),
    "ekf_get_motion_jacobian": (C.c_int, [_P, _P, _P]),
    "ekf_get_predictions": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "ekf_update": (C.c_int, [_P, _P, _P, C.c_int, C.c_int]),
    "ekf_update_device": (C.c_int, [_P, _P, _P, C.c_int, C.c_int]),
    "ekf_rescue_high_innovation": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_double, _P]),
    "ekf_set_frame": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int]),
    "ekf_set_patch": (C.c_int, [_P, C.c_int, _P]),
    "ekf_get_patch": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "ekf_get_blur_predictions": (C.c_int, [_P, _P]),
    "ekf_find_matches": (C.c_int, [_P, C.c_double, _P, _P, _P]),
    "ekf_export_points": (C.c_int, [_P, _P, C.c_int]),
    "ekf_export_points_table": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "ekf_get_feature_ids": (C.c_int, [_P, _P, _P]),
    "ekf_set_feature_meta": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "ekf_num_archived": (C.c_int, [_P]),
    "ekf_get_search_ellipses": (C.c_int, [_P, C.c_int, _P]),
    "ekf_ransac_1point": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, _P, _P, C.POINTER(C.c_int)]),
    "ekf_update_two_stage": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_uint, C.c_double, C.c_double, _P, _P,
                                       C.POINTER(C.c_int)]),
    "ekf_innovation_covariance": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "ekf_get_gain": (C.c_int, [_P, _P]),
    "ekf_last_measurement_rows": (C.c_int, [_P]),
    "ekf_convert_xyz_if_linear": (C.c_int, [_P, C.c_int]),
    "ekf_convert_xyz_if_linear_all": (C.c_int, [_P]),
    "ekf_num_features": (C.c_int, [_P]),
    "ekf_state_dim": (C.c_int, [_P]),
    "ekf_get_feature_layout": (C.c_int, [_P, _P, _P]),
    "ekf_get_state": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "ekf_set_state": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "ekf_get_sigma_block": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ekf_peek_workspace": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ekf_set_sigma_block": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ekf_covariance_parameter": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "ekf_feature_xyz": (C.c_int, [_P, C.c_int, _P, _P]),
    "ekf_check_invariants": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ekf_profile_kernels": (C.c_int, []),
    "ekf_profile_kernel_name": (C.c_char_p, [C.c_int]),
    "ekf_profile_read": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "ekf_profile_reset": (C.c_int, [_P]),
    "ekf_profile_work": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double)]),
    "ekf_get_chunk_plan": (C.c_int, [_P, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ekf_launch_kinds": (C.c_int, []),
    "ekf_launch_kind_name": (C.c_char_p, [C.c_int]),
    "ekf_launch_count": (C.c_int, [_P, C.c_int, C.POINTER(C.c_longlong)]),
    "ekf_shard_configure": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "ekf_shard_get_info": (C.c_int, [_P, _P]),
    "ekf_shard_update": (C.c_int, [_P, _P, _P, C.c_int, C.c_int]),
    "ekf_shard_rebalance": (C.c_int, [_P]),
    "ekf_device_mu": (_P, [_P]),
    "ekf_device_sigma": (_P, [_P, C.POINTER(C.c_int)]),
}

_lib = None


def load_library(path: str = LIB_PATH):
    """dlopen the HIP library and attach prototypes.  Fails loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise EkfError(3, f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
    lib = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)            # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
