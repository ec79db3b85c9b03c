"""MI355X-native EKF-MonoSLAM predict/update core (host-side Python mirror).

The product is `lib/libekfslam_hip.so` (hand-written HIP for gfx950 behind the C ABI of
`include/ekf_monoslam.h`).  This package only binds it: `capi` is the ctypes layer,
`VSlamFilter` mirrors the public math methods of the reference's `class VSlamFilter`
(mono-slam/src/vslamRansac.hpp:94-141).  There is no CPU fallback: importing works
anywhere, constructing a filter needs the built library and a HIP device.

The directory name contains '-', so it is imported through `__graft_entry__.load_package()`
(alias `ekf_monoslam_amd`).
"""
from .capi import (EkfConfig, EkfError, LIB_PATH, declared_symbols, load_library)  # noqa: F401
from .vslam_filter import VSlamFilter, kinect_config, sim_config  # noqa: F401
