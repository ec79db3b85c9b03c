"""Image side of the HIP path against the oracle: template capture, predicted blur,
NCC search -- all through the C ABI (ekf_set_frame / ekf_find_matches / ...)."""
import dataclasses

import numpy as np
import pytest

import ekf_oracle as o
import image_oracle as io_
from __graft_entry__ import load_package
from helpers import bound, relf

pytestmark = pytest.mark.gpu


def _pair(n_feat, dtype, frame, w=(0.0, 0.05, 0.0), kernel_size=1000, t_camera=0.0, capacity=None):
    """Oracle filter + HIP filter with the same features; templates captured from `frame` on both sides."""
    pkg = load_package()
    cfg = dataclasses.replace(o.Config.kinect(), kernel_size=kernel_size, T_camera=t_camera)
    ref = o.build_scenario(o.StructuredFilter, cfg, n_feat, dtype, w=w)
    gcfg = dict(pkg.kinect_config())
    gcfg.update(kernel_size=kernel_size, T_camera=t_camera)
    g = pkg.VSlamFilter(gcfg, capacity_features=capacity or n_feat, dtype=dtype)
    g.setDt(ref.dT)
    full = g.getFullState()
    full[7:13] = ref.mu[7:13]
    g.setFullState(full)
    g.setFrame(frame)
    px = o.synthetic_pixels(cfg, n_feat)
    for (u, v) in px:
        assert g.addFeature((u, v)) == 1
    g.setFullState(ref.mu)
    g.setSigmaBlock(ref.Sigma)
    tpl = [io_.capture_patch(frame, u, v, cfg.window_size) for (u, v) in px]
    return ref, g, tpl, cfg


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_template_capture_and_removal(dtype):
    frame = io_.random_texture(240, 320, seed=21)
    ref, g, tpl, cfg = _pair(12, dtype, frame)
    for i in range(12):
        assert np.array_equal(g.getPatch(i), tpl[i])
        assert np.array_equal(g.getPatch(i, matching=True), tpl[i])
    g.removeFeatures([2, 7])                                   # the templates follow their features
    keep = [i for i in range(12) if i not in (2, 7)]
    for k, i in enumerate(keep):
        assert np.array_equal(g.getPatch(k), tpl[i])
    custom = (np.arange(cfg.window_size ** 2) % 251).astype(np.uint8).reshape(cfg.window_size, -1)
    g.setPatch(3, custom)
    assert np.array_equal(g.getPatch(3), custom)
    with pytest.raises(Exception):
        g.getPatch(10)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_predicted_blur(dtype):                                # vR.cpp:496-500, 546-548; libblur.cpp:17-79
    frame = io_.random_texture(240, 320, seed=22)
    # fast rotation + translation: the blur line is several pixels long and oblique
    ref, g, tpl, cfg = _pair(24, dtype, frame, w=(0.6, 1.6, -0.4), kernel_size=2, t_camera=0.5)
    ref.predict()
    g.predict()
    hb = g.blurPredictions()
    h, vis, rem, S2 = g.predictions()
    n_blurred = 0
    for i, ft in enumerate(ref.features):
        hb_ref = io_.blur_point(ref, ft)
        tol = 2e-3 if dtype == np.float32 else 1e-8
        assert np.allclose(hb[i], hb_ref, rtol=0, atol=tol)
        if not ft.is_in_innovation:
            continue
        # the template arithmetic is float/double on both sides: bit-exact for the same (h, hb)
        want = io_.matching_patch(tpl[i], h[i], hb[i], cfg.kernel_size)
        got = g.getPatch(i, matching=True)
        assert np.array_equal(got, want), i
        assert np.array_equal(g.getPatch(i), tpl[i])           # Patch::patch itself never changes
        n_blurred += int(not np.array_equal(want, tpl[i]))
    assert n_blurred >= 10


def test_no_blur_below_kernel_size():
    frame = io_.random_texture(240, 320, seed=23)
    ref, g, tpl, cfg = _pair(8, np.float32, frame, kernel_size=1000, t_camera=0.5)
    g.predict()
    for i in range(8):
        assert np.array_equal(g.getPatch(i, matching=True), tpl[i])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ncc_search_matches_oracle(dtype):                     # Patch::findMatch, Patch.cpp:215-293
    frame = io_.random_texture(240, 320, seed=24)
    ref, g, tpl, cfg = _pair(40, dtype, frame)
    ref.predict()
    g.predict()
    moved = np.roll(np.roll(frame, 2, axis=1), -1, axis=0)      # the scene moved +2, -1 pixels
    px = o.synthetic_pixels(cfg, 40)
    for k in (0, 1, 2):                                         # flat occluders hide three of the features
        u, v = int(px[k][0]) + 2, int(px[k][1]) - 1
        moved[max(v - 30, 0):v + 30, max(u - 30, 0):u + 30] = 128
    g.setFrame(moved)
    z, found, score = g.findMatches()
    h, vis, rem, S2 = g.predictions()
    n_found = 0
    for i, ft in enumerate(ref.features):
        if not ft.is_in_innovation:
            assert not found[i] and score[i] == -1
            continue
        # the search consumes the DEVICE predictions (h, 2x2 St block) exactly as the reference consumes its own
        S = S2[i]
        ok, zz, sc, win = io_.find_match(moved, tpl[i], h[i], S, cfg.sigma_size)
        assert bool(found[i]) == ok, i
        assert tuple(int(v) for v in z[i]) == tuple(zz), i
        if np.isfinite(sc):
            assert abs(float(score[i]) - float(sc)) < 1e-6
        if ok:
            n_found += 1
            assert np.array_equal(g.getPatch(i, matching=True), win)     # Patch.cpp:286
    assert n_found >= 25 and not found[0] and not found[1] and not found[2]
    # device predictions vs the oracle's own, so that the comparison above is anchored
    for i, ft in enumerate(ref.features):
        assert np.allclose(h[i], ft.h, atol=2e-3 if dtype == np.float32 else 1e-9)


def test_matched_measurements_drive_the_update():
    """predict -> findMatches -> update with the matched set: the full reference frame loop on the device."""
    frame = io_.random_texture(240, 320, seed=25)
    ref, g, tpl, cfg = _pair(30, np.float32, frame)
    ref.predict()
    g.predict()
    z, found, score = g.findMatches()                          # same frame: every visible template is found in place
    idx = [i for i in range(30) if found[i]]
    assert len(idx) >= 25 and np.all(score[idx] > 0.999)
    zz = z[idx].reshape(-1)
    ref.update(zz.astype(ref.T), idx)
    g.update(zz, idx)
    assert bound("g.getFullState(), ref.mu", relf(g.getFullState(), ref.mu), 2e-5) and bound("g.getFullSigma(), ref.Sigma", relf(g.getFullSigma(), ref.Sigma), 5e-4)


def test_image_error_paths():
    pkg = load_package()
    g = pkg.VSlamFilter(pkg.kinect_config(), capacity_features=4)
    with pytest.raises(Exception):
        g.findMatches()                                         # no frame
    with pytest.raises(Exception):
        g.setFrame(np.zeros((100, 100), np.uint8))              # wrong size
    g.setFrame(np.zeros((240, 320), np.uint8))
    with pytest.raises(Exception):
        g.findMatches()                                         # no predictions yet
    g.predict()
    z, found, score = g.findMatches()                           # empty map
    assert z.shape == (0, 2) and found.size == 0


@pytest.mark.parametrize("window,cfg_name", [(30, "sim"), (21, "kinect"), (32, "kinect"), (9, "kinect")])
def test_other_window_sizes(window, cfg_name):
    """Template edges that are even, not a multiple of 4, the largest supported and a small one (conf_sim.cfg has
    window_size = 30, sigma_size = 4; ConfigVSLAM's default is 21): capture, blur and search against the oracle."""
    pkg = load_package()
    base_o = o.Config.sim() if cfg_name == "sim" else o.Config.kinect()
    cfg = dataclasses.replace(base_o, window_size=window, kernel_size=2, T_camera=0.5)
    gcfg = dict(pkg.sim_config() if cfg_name == "sim" else pkg.kinect_config())
    gcfg.update(window_size=window, kernel_size=2, T_camera=0.5)
    n_feat = 16
    frame = io_.random_texture(cfg.image_height, cfg.image_width, seed=40 + window)
    ref = o.build_scenario(o.StructuredFilter, cfg, n_feat, np.float32, w=(0.3, 0.9, -0.2))
    g = pkg.VSlamFilter(gcfg, capacity_features=n_feat, dtype=np.float32)
    g.setDt(ref.dT)
    full = g.getFullState()
    full[7:13] = ref.mu[7:13]
    g.setFullState(full)
    g.setFrame(frame)
    px = o.synthetic_pixels(cfg, n_feat)
    for (u, v) in px:
        assert g.addFeature((u, v)) == 1
    g.setFullState(ref.mu)
    g.setSigmaBlock(ref.Sigma)
    tpl = [io_.capture_patch(frame, u, v, window) for (u, v) in px]
    for i in range(n_feat):
        assert np.array_equal(g.getPatch(i), tpl[i])
    ref.predict()
    g.predict()
    h, vis, rem, S2 = g.predictions()
    hb = g.blurPredictions()
    for i, ft in enumerate(ref.features):
        if ft.is_in_innovation:
            assert np.array_equal(g.getPatch(i, matching=True), io_.matching_patch(tpl[i], h[i], hb[i], cfg.kernel_size)), i
    moved = np.roll(np.roll(frame, -1, axis=1), 2, axis=0)
    g.setFrame(moved)
    z, found, score = g.findMatches()
    checked = 0
    for i, ft in enumerate(ref.features):
        if not ft.is_in_innovation:
            continue
        mp_ = io_.matching_patch(tpl[i], h[i], hb[i], cfg.kernel_size)
        ok, zz, sc, win = io_.find_match(moved, mp_, h[i], S2[i], cfg.sigma_size)
        assert bool(found[i]) == ok and tuple(int(v) for v in z[i]) == tuple(zz), i
        if np.isfinite(sc):
            assert abs(float(score[i]) - float(sc)) < 1e-6
        checked += 1
    assert checked >= 8
