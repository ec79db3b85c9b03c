"""CPU-side checks of the drop-in boundary: the library builds, loads, and exports every
symbol include/ekf_monoslam.h declares; no compute call is made (no GPU here)."""
import ctypes as C
import os

import numpy as np
import pytest

import __graft_entry__ as entry


@pytest.fixture(scope="module")
def pkg():
    entry.build()
    return entry.load_package()


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    names = pkg.declared_symbols()
    assert len(names) >= 35
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/ekf_monoslam.h but not exported"
    assert lib.ekf_abi_version() == 6


def test_prototypes_cover_the_header(pkg):
    from ekf_monoslam_amd import capi
    assert sorted(capi._PROTOS) == pkg.declared_symbols()


def test_config_default_matches_reference_defaults(pkg):
    lib = pkg.load_library()
    c = pkg.EkfConfig()
    lib.ekf_config_default(C.byref(c))
    # ConfigVSLAM.cpp:27-47
    assert np.isclose(c.sigma_vx, 0.01) and np.isclose(c.sigma_wz, 0.01)
    assert (c.window_size, c.sigma_pixel, c.scale, c.sigma_size) == (21, 2, 1, 2)
    assert np.isclose(c.rho_0, 0.1) and np.isclose(c.sigma_rho_0, 0.25) and np.isclose(c.T_camera, 0.5)
    assert (c.nInitFeatures, c.min_features, c.max_features, c.forsePlane) == (5, 30, 100, 0)
    # camModel.hpp:22-31
    assert np.isclose(c.fx, 592.2860) and np.isclose(c.k2, 0.5521) and np.isclose(c.p2, 0.0140)


def test_oracle_and_binding_parameter_sets_agree(pkg):
    import ekf_oracle as o
    for ours, theirs in ((pkg.kinect_config(), o.Config.kinect()), (pkg.sim_config(), o.Config.sim())):
        for k, v in ours.items():
            assert np.isclose(v, getattr(theirs, k)), k


def test_create_fails_loudly_without_a_device(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.EkfError) as ei:
        pkg.VSlamFilter(pkg.kinect_config())
    assert "no CPU fallback" in str(ei.value)


def test_bad_arguments_are_rejected_before_touching_the_device(pkg):
    lib = pkg.load_library()
    h = C.c_void_p()
    c = pkg.EkfConfig()
    lib.ekf_config_default(C.byref(c))
    assert lib.ekf_create(C.byref(c), 12, 10, 0, 0, C.byref(h)) == 1      # camera_dim
    assert lib.ekf_create(C.byref(c), 14, 10, 7, 0, C.byref(h)) == 1      # dtype
    assert lib.ekf_predict(None, None, None, 0) == 1
    assert lib.ekf_num_features(None) == 0


def test_missing_library_is_an_error_not_a_fallback(pkg, tmp_path):
    from ekf_monoslam_amd import capi
    saved = capi._lib
    capi._lib = None
    try:
        with pytest.raises(pkg.EkfError):
            capi.load_library(str(tmp_path / "nope.so"))
    finally:
        capi._lib = saved
