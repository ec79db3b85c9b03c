"""Known-answer tests of the image-side oracle (Patch.cpp / libblur.cpp restatement)."""
import numpy as np

import image_oracle as io_


def test_correlation_known_answers():                          # computeCorrelation, Patch.cpp:295-329
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (15, 15)).astype(np.uint8)
    assert io_.compute_correlation(a, a) == np.float32(1.0)
    assert io_.compute_correlation(a, 255 - a) == np.float32(-1.0)
    assert abs(io_.compute_correlation(a, (a // 2 + 10).astype(np.uint8)) - 1.0) < 2e-3     # affine gain / offset
    assert np.isnan(io_.compute_correlation(a, np.full_like(a, 7)))                         # zero variance: 0 / 0
    b = rng.integers(0, 256, (15, 15)).astype(np.uint8)
    c = io_.compute_correlation(a, b)
    assert abs(c - np.corrcoef(a.ravel().astype(float), b.ravel().astype(float))[0, 1]) < 1e-6


def test_line_kernel_known_answers():                          # evaluateKernel, libblur.cpp:17-52
    k = io_.evaluate_kernel((10.0, 5.0), (6.0, 5.0))           # 4 px along +x: 1 x 5 kernel, 4 cells of 1/4
    assert k.shape == (1, 5) and np.allclose(k, [[0.25, 0.25, 0.25, 0.25, 0.0]])
    k = io_.evaluate_kernel((6.0, 5.0), (10.0, 5.0))           # the other way: theta = pi (float), y0 = int(0.999.. * 4) = 3: the same cells
    assert k.shape == (1, 5) and np.allclose(k, [[0.25, 0.25, 0.25, 0.25, 0.0]])
    k = io_.evaluate_kernel((5.0, 5.0), (5.0, 9.0))            # along -y: 5 x 1
    assert k.shape == (5, 1) and abs(k.sum() - 1) < 1e-15
    k = io_.evaluate_kernel((3.0, 3.0), (0.0, 0.0))            # diagonal: distinct cells on the diagonal
    assert k.shape == (4, 4) and np.count_nonzero(k - np.diag(np.diag(k))) == 0


def test_filter2d_and_rounding():                              # filter2D / convertTo, libblur.cpp:73-76
    rng = np.random.default_rng(4)
    p = rng.integers(0, 256, (7, 7)).astype(np.float64)
    delta = np.zeros((3, 3))
    delta[1, 1] = 1.0
    assert np.array_equal(io_.filter2d_reflect101(p, delta), p)
    shift = np.zeros((1, 3))
    shift[0, 2] = 1.0                                           # dst(x) = src(x + 1), reflect-101 at the border
    out = io_.filter2d_reflect101(p, shift)
    assert np.array_equal(out[:, :-1], p[:, 1:]) and np.array_equal(out[:, -1], p[:, -2])
    assert [io_._reflect101(q, 5) for q in (-2, -1, 0, 4, 5, 6, 9)] == [2, 1, 0, 4, 3, 2, 1]
    assert io_.to_u8(np.array([0.5, 1.5, 2.5, -3.0, 300.0])).tolist() == [0, 2, 2, 0, 255]


def test_lu_inverse_matches_linalg():
    rng = np.random.default_rng(6)
    for _ in range(20):
        a = rng.normal(size=(2, 2)).astype(np.float32)
        s = (a @ a.T + np.eye(2, dtype=np.float32) * 4).astype(np.float32)
        if rng.random() < 0.5:
            s = s[::-1].copy()                                  # force the pivot swap
        assert np.allclose(io_.lu_inverse_2x2(s), np.linalg.inv(s.astype(np.float64)), rtol=2e-5, atol=1e-6)


def test_find_match_recovers_a_known_shift():                  # Patch::findMatch, Patch.cpp:215-293
    img = io_.random_texture(120, 160, seed=11)
    w = 15
    tpl = io_.capture_patch(img, 80.4, 60.7, w)                 # centred on (80, 60)
    moved = np.roll(np.roll(img, 3, axis=1), -2, axis=0)        # content moves +3 in u, -2 in v
    S = np.array([[9.0, 1.0], [1.0, 6.0]], np.float32)
    found, z, score, win = io_.find_match(moved, tpl, (81.2, 60.1), S, 2)
    assert found and z == (83, 58) and score == np.float32(1.0) and np.array_equal(win, tpl)
    # outside the ellipse (tiny covariance): the true position is not a candidate
    found2, z2, score2, _ = io_.find_match(moved, tpl, (81.2, 60.1), np.eye(2, dtype=np.float32) * 0.25, 2)
    assert not found2 and z2 == (-1, -1) and score2 < 0.8
    # a prediction whose whole window is off the searchable area: no candidate at all
    found3, _, score3, _ = io_.find_match(moved, tpl, (3.0, 3.0), S, 2)
    assert not found3 and score3 == np.float32(-1)


def test_image_oracle_reproduces_golden():
    """tests/golden/image_w15.npz (make_golden_image.py): regression guard on the image oracle."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "image_w15.npz"))
    frame, moved = g["frame"], g["moved"]
    assert np.array_equal(io_.random_texture(96, 128, seed=31), frame)
    for i in range(6):
        tpl = io_.capture_patch(frame, g["centres"][i][0], g["centres"][i][1], 15)
        assert np.array_equal(tpl, g["templates"][i])
        assert np.array_equal(io_.matching_patch(tpl, g["blur_h"][i], g["blur_hb"][i], 2), g["blurred"][i])
        ok, z, sc, _ = io_.find_match(moved, tpl, g["h_pred"][i], g["S"][i], 2)
        assert ok == bool(g["found"][i]) and tuple(z) == tuple(g["z"][i]) and sc == g["score"][i]
    assert np.array_equal(g["blurred"][0], g["templates"][0])            # motion shorter than kernel_size: a copy
    assert not np.array_equal(g["blurred"][3], g["templates"][3])
