"""CPU restatement of the image side of the reference filter -- TEST INFRASTRUCTURE ONLY.

**PARITY UNPINNED**: the reference holds no tests or fixtures for this path and neither it nor its
third-party arithmetic (OpenCV `filter2D` / `convertTo`, Eigen `MatrixXf::inverse()`) can be built
here (no OpenCV, no Eigen in the image).  The OpenCV / Eigen pieces are restated from
their published algorithms:
  * cv::filter2D: correlation, anchor (-1,-1) = kernel centre (ksize/2), BORDER_DEFAULT =
    BORDER_REFLECT_101, accumulation in double over the non-zero kernel cells in (row, col) order;
  * Mat::convertTo(CV_8U) from double: cvRound (round half to even) then saturation;
  * Eigen MatrixXf::inverse() of a run-time sized matrix: PartialPivLU, inverse = solve(Identity)
    with unit-lower and upper triangular solves that multiply by the reciprocal of the pivot.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows (file:line of /root/reference/mono-slam/src):
  Patch::Patch template capture      vslamRansac.cpp:318, Patch.cpp:76-105
  blur pose and Patch::blur          vslamRansac.cpp:496-500, 546-548, 575-576; Patch.cpp:50-57
  evaluateKernel / blurPatch         libblur.cpp:17-52, 62-87
  Patch::findMatch                   Patch.cpp:215-293
  computeCorrelation                 Patch.cpp:295-329
"""
from __future__ import annotations

import math

import numpy as np

try:                                      # tests put oracle/ on sys.path and import the modules top-level
    import ekf_oracle as o
except ImportError:                       # package-style import (bench.py, smoke)
    from . import ekf_oracle as o

F = np.float32
PATCH_MATCHING_THRESHOLD = 0.8            # Patch.cpp:14


def capture_patch(frame, u, v, window):
    """cv::Mat(frame, cv::Rect(pf.x - w/2, pf.y - w/2, w, w)).clone(): float -> int truncation."""
    x0 = int(F(u) - F(window // 2))
    y0 = int(F(v) - F(window // 2))
    return np.array(frame[y0:y0 + window, x0:x0 + window], dtype=np.uint8, copy=True)


def blur_point(filt, ft):
    """hi_out_blurred (vR.cpp:496-499, 546 / 575): prediction of the feature at the blur pose."""
    T = filt.T
    mu = filt.mu
    tc = T(filt.cfg.T_camera)
    dT = T(filt.dT)
    r, q, v, w = mu[0:3], mu[3:7], mu[7:10], mu[10:13]
    RotCW_b = o.quat2rot(o.quat_complement(o.quat_product(q, o.vec2quat(w * tc * dT, T), T), T), T)
    r_b = r + v * tc * dT
    pos = ft.position_in_state
    if ft.coding == o.INV:
        d, _ = o.inverse2xyz_projecting(mu[pos:pos + 6], r_b, T, False)
    else:
        d = mu[pos:pos + 3] - r_b
    hb, _ = filt.cam.project(RotCW_b @ d, False)
    return hb


def evaluate_kernel(one, two):
    """libblur.cpp:17-52.  one, two: (x, y) float32 points.  Returns the normalised line kernel (double)."""
    ox, oy, tx, ty = F(one[0]), F(one[1]), F(two[0]), F(two[1])
    dx, dy = F(ox - tx), F(oy - ty)
    height = int(F(abs(dx)) + F(1))                   # kernel columns
    width = int(F(abs(dy)) + F(1))                    # kernel rows
    kernel = np.zeros((width, height), np.float64)
    theta = float(np.arctan2(dy, dx, dtype=F))        # atan2(float, float) -> float, widened to double
    length = math.sqrt(float(dx) * float(dx) + float(dy) * float(dy))     # cv::norm(Point2f): double
    c, s = math.cos(theta), math.sin(theta)
    x0 = int(-s * length) if s < 0 else 0
    y0 = int(-c * length) if c < 0 else 0
    i = 0
    while i < length:
        x = int(i * s + x0)
        y = int(i * c + y0)
        kernel[min(max(x, 0), width - 1), min(max(y, 0), height - 1)] = 1.0
        i += 1
    return kernel / kernel.sum()


def _reflect101(p, n):
    if n == 1:
        return 0
    while p < 0 or p >= n:
        p = -p if p < 0 else 2 * (n - 1) - p
    return p


def filter2d_reflect101(src, kernel):
    """cv::filter2D(src, dst, -1, kernel, Point(-1,-1), 0, BORDER_DEFAULT) on a CV_64F image."""
    src = np.asarray(src, np.float64)
    rows, cols = src.shape
    kr, kc = kernel.shape
    ay, ax = kr // 2, kc // 2
    taps = [(r, c, kernel[r, c]) for r in range(kr) for c in range(kc) if kernel[r, c] != 0.0]
    dst = np.zeros_like(src)
    ys = np.arange(rows)
    xs = np.arange(cols)
    for r, c, kv in taps:                              # (row, col) order, one accumulation per cell
        sy = np.array([_reflect101(int(y) + r - ay, rows) for y in ys])
        sx = np.array([_reflect101(int(x) + c - ax, cols) for x in xs])
        dst = dst + kv * src[np.ix_(sy, sx)]
    return dst


def to_u8(img):
    """Mat::convertTo(CV_8U): cvRound (half to even) + saturate."""
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def blur_patch(patch, one, two):
    """blurPatch, libblur.cpp:62-87."""
    return to_u8(filter2d_reflect101(np.asarray(patch, np.float64), evaluate_kernel(one, two)))


def matching_patch(patch, h, hb, kernel_min_size):
    """Patch::blur, Patch.cpp:50-57 ((p1 - p2).norm() in float)."""
    dx, dy = F(F(h[0]) - F(hb[0])), F(F(h[1]) - F(hb[1]))
    nrm = np.sqrt(F(F(dx * dx) + F(dy * dy)), dtype=F)
    if nrm > F(kernel_min_size):
        return blur_patch(patch, (F(h[0]), F(h[1])), (F(hb[0]), F(hb[1])))
    return np.array(patch, dtype=np.uint8, copy=True)


def compute_correlation(f1, f2):
    """computeCorrelation, Patch.cpp:295-329: double sums, float result."""
    a = np.asarray(f1, np.float64)
    b = np.asarray(f2, np.float64)
    n = a.size
    m1 = a.sum() / n
    m2 = b.sum() / n
    n1 = ((a - m1) * (a - m1)).sum()
    n2 = ((b - m2) * (b - m2)).sum()
    corr = ((a - m1) * (b - m2)).sum()
    with np.errstate(invalid="ignore", divide="ignore"):
        return F(np.float64(corr) / np.sqrt(np.float64(n2 * n1)))


def lu_inverse_2x2(S):
    """MatrixXf::inverse() of a dynamic 2x2: PartialPivLU + solve(Identity), float32 throughout."""
    s00, s01, s10, s11 = F(S[0, 0]), F(S[0, 1]), F(S[1, 0]), F(S[1, 1])
    swap = abs(s10) > abs(s00)
    p, q, r, s = (s10, s11, s00, s01) if swap else (s00, s01, s10, s11)
    l = F(r / p)
    u11 = F(s - F(l * q))
    ip, iu = F(F(1) / p), F(F(1) / u11)
    inv = np.zeros((2, 2), F)
    for c in range(2):
        b0 = F(1.0 if ((c == 1) if swap else (c == 0)) else 0.0)
        b1 = F(1.0 if ((c == 0) if swap else (c == 1)) else 0.0)
        y1 = F(b1 - F(l * b0))
        x1 = F(y1 * iu)
        x0 = F(F(b0 - F(q * x1)) * ip)
        inv[0, c], inv[1, c] = x0, x1
    return inv


def find_match(frame, mpatch, h, S, sigma_size, threshold=PATCH_MATCHING_THRESHOLD):
    """Patch::findMatch, Patch.cpp:215-293.  Returns (found, (zu, zv), score, matched_window or None)."""
    frame = np.asarray(frame, np.uint8)
    fh, fw = frame.shape
    w = mpatch.shape[1]
    hw = w // 2
    uc, vc = int(F(h[0])), int(F(h[1]))
    S = np.asarray(S, F)
    inv = lu_inverse_2x2(S)
    x2, y2, yx = inv[0, 0], inv[1, 1], F(F(2) * inv[1, 0])
    mx = F(-1)
    sig = F(sigma_size)
    sigma_2 = F(sig * sig)
    du = F(np.float64(sig) * np.sqrt(np.float64(S[0, 0])))
    dv = F(np.float64(sig) * np.sqrt(np.float64(S[1, 1])))
    if du > 20:
        du = F(20)
    if dv > 20:
        dv = F(20)
    center, new_patch = (-1, -1), None
    i = int(F(F(uc) - du))
    while F(i) <= F(F(uc) + du):
        j = int(F(F(vc) - dv))
        while F(j) <= F(F(vc) + dv):
            if i > hw and j > hw and i < fw - hw and j < fh - hw:
                fi, fj = F(i - uc), F(j - vc)
                g = F(F(F(F(x2 * fi) * fi) + F(F(y2 * fj) * fj)) + F(F(yx * fi) * fj))
                if g <= sigma_2:
                    sub = frame[j - hw:j - hw + w, i - hw:i - hw + w]
                    val = compute_correlation(mpatch, sub)
                    if val > mx:
                        center, mx, new_patch = (i, j), val, sub.copy()
            j += 1
        i += 1
    if mx < F(threshold):
        return False, (-1, -1), (mx if new_patch is not None else F(-1)), None
    return True, center, mx, new_patch


# --------------------------------------------------------------------------------------------
# synthetic imagery for the tests: a textured plane rendered by shifting a random field
# --------------------------------------------------------------------------------------------
def random_texture(height, width, seed, smooth=2):
    """Band-limited random 8-bit image (box-filtered white noise): distinctive at patch scale."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(height + 2 * smooth, width + 2 * smooth)).astype(np.float64)
    acc = np.zeros((height, width))
    for dy in range(2 * smooth + 1):
        for dx in range(2 * smooth + 1):
            acc += img[dy:dy + height, dx:dx + width]
    acc /= (2 * smooth + 1) ** 2
    acc = (acc - acc.min()) / (acc.max() - acc.min()) * 255.0
    return to_u8(acc)
