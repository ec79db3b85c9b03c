"""The exact sharded-vs-plain comparison (helpers.exact_or_anchor_glitch) is STRICT: the one classified deviation of
DESIGN 6 fails too, with its own message; only EKF_ALLOW_ANCHOR_GLITCH=1 downgrades that signature to a warning."""
import warnings

import numpy as np
import pytest

from helpers import exact_or_anchor_glitch


def _state(n_feat=50, seed=3):
    rng = np.random.default_rng(seed)
    n = 14 + 6 * n_feat
    mu = rng.standard_normal(n).astype(np.float32)
    for f in range(n_feat):
        mu[14 + 6 * f:14 + 6 * f + 3] = rng.standard_normal(3).astype(np.float32) * 1e-15     # anchors of a run from the origin
    rows = np.r_[0:14, 100:140]
    S = (rng.standard_normal((len(rows), n)) * 1e-4).astype(np.float32)
    return mu, rows, S


def test_bit_identical_passes_silently():
    mu, rows, S = _state()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert exact_or_anchor_glitch("x", mu, mu.copy(), S, S.copy(), rows) == ""


def test_one_anchor_coordinate_fails_by_default_and_is_classified(monkeypatch):
    monkeypatch.delenv("EKF_ALLOW_ANCHOR_GLITCH", raising=False)
    mu, rows, S = _state()
    mu2, S2 = mu.copy(), S.copy()
    i = 14 + 6 * 17 + 1
    mu2[i] = np.float32(-4.26e-9)
    S2[:, i] += np.float32(3e-10)
    with pytest.raises(AssertionError, match="anchor-coordinate glitch"):
        exact_or_anchor_glitch("x", mu2, mu, S2, S, rows)


def test_one_anchor_coordinate_is_a_warning_only_when_allowed(monkeypatch):
    monkeypatch.setenv("EKF_ALLOW_ANCHOR_GLITCH", "1")
    mu, rows, S = _state()
    mu2, S2 = mu.copy(), S.copy()
    i = 14 + 6 * 17 + 1                                     # y_a of feature 17
    mu2[i] = np.float32(-4.26e-9)
    S2[:, i] += np.float32(3e-10)
    mu2[14 + 6 * 3] = np.nextafter(mu2[14 + 6 * 3], np.float32(1))   # the rounding-level echo in another anchor row
    with pytest.warns(UserWarning, match="anchor-coordinate glitch"):
        msg = exact_or_anchor_glitch("x", mu2, mu, S2, S, rows)
    assert "entries of mu differ" in msg


@pytest.mark.parametrize("case", ["angle", "large", "many", "sigma"])
def test_anything_else_fails(case):
    mu, rows, S = _state()
    mu2, S2 = mu.copy(), S.copy()
    if case == "angle":
        mu2[14 + 6 * 17 + 3] = np.nextafter(mu2[14 + 6 * 17 + 3], np.float32(9))     # theta: not an anchor coordinate
    elif case == "large":
        mu2[14 + 6 * 17 + 1] = np.float32(1e-6)
    elif case == "many":
        for f in (1, 2, 3, 4):
            mu2[14 + 6 * f + 1] = np.float32(1e-9)
    else:
        S2[3, 200] += np.float32(1e-6)
    with pytest.raises(AssertionError):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            exact_or_anchor_glitch("x", mu2, mu, S2, S, rows)
