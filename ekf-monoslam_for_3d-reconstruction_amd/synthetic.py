"""Synthetic N-feature measurement streams for bench.py and the stream tests.

Pure numpy scenario generation -- not part of the filter: a static cloud of world points seen
by a camera on a small periodic trajectory, projected with the pinhole + radial/tangential
model of the configured camera, plus Gaussian pixel noise.  Seeds: 1234 pixels/depths,
1235 noise.
"""
from __future__ import annotations

import numpy as np

Q0 = np.array([0.0, 0.0, -0.707106781, 0.707106781])       # initial attitude, vR.cpp:180
AMP = 1.0                  # trajectory amplitude scale
DEPTH = (2.0, 10.0)        # true feature depth range [m]


def quat2rot(q):
    r, i, j, k = q
    return np.array([
        [r*r + i*i - j*j - k*k, 2*(i*j - r*k), 2*(r*j + i*k)],
        [2*(r*k + i*j), r*r - i*i + j*j - k*k, 2*(j*k - r*i)],
        [2*(i*k - r*j), 2*(r*i + j*k), r*r - i*i - j*j + k*k]])


def quat_mul(a, b):
    return np.array([a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3],
                     a[1]*b[0] + a[0]*b[1] - a[3]*b[2] + a[2]*b[3],
                     a[2]*b[0] + a[3]*b[1] + a[0]*b[2] - a[1]*b[3],
                     a[3]*b[0] - a[2]*b[1] + a[1]*b[2] + a[0]*b[3]])


def project(cfg, pc):
    """pc: (..., 3) camera-frame points -> (..., 2) distorted pixels."""
    x1 = pc[..., 0] / pc[..., 2]
    y1 = pc[..., 1] / pc[..., 2]
    r2 = x1*x1 + y1*y1
    l = 1 + cfg["k1"]*r2 + cfg["k2"]*r2*r2 + cfg["k3"]*r2**3
    x2 = x1*l + 2*cfg["p1"]*x1*y1 + cfg["p2"]*(r2 + 2*x1*x1)
    y2 = y1*l + 2*cfg["p2"]*x1*y1 + cfg["p1"]*(r2 + 2*y1*y1)
    return np.stack([cfg["fx"]*x2 + cfg["u0"], cfg["fy"]*y2 + cfg["v0"]], axis=-1)


def unproject(cfg, px):
    x2 = (px[..., 0] - cfg["u0"]) / cfg["fx"]
    y2 = (px[..., 1] - cfg["v0"]) / cfg["fy"]
    x1, y1 = x2.copy(), y2.copy()
    for _ in range(50):
        r2 = x1*x1 + y1*y1
        l = 1 + cfg["k1"]*r2 + cfg["k2"]*r2*r2 + cfg["k3"]*r2**3
        dx = 2*cfg["p1"]*x1*y1 + cfg["p2"]*(r2 + 2*x1*x1)
        dy = 2*cfg["p2"]*x1*y1 + cfg["p1"]*(r2 + 2*y1*y1)
        x1 = (x2 - dx) / l
        y1 = (y2 - dy) / l
    return np.stack([x1, y1, np.ones_like(x1)], axis=-1)


def initial_pixels(cfg, n_features, seed=1234, margin=45):
    rng = np.random.default_rng(seed)
    u = rng.uniform(margin, cfg["image_width"] - margin, size=n_features)
    v = rng.uniform(margin, cfg["image_height"] - margin, size=n_features)
    return np.stack([u, v], axis=1)


def trajectory(t):
    """True camera pose at time t: small periodic motion so every feature stays in view."""
    r = AMP * np.array([0.15 * np.sin(2*np.pi*t/2.0), 0.05 * np.sin(2*np.pi*t/3.0),
                        0.10 * (1.0 - np.cos(2*np.pi*t/4.0))])
    ang = 0.05 * np.sin(2*np.pi*t/2.5)
    dq = np.array([np.cos(ang/2), 0.0, np.sin(ang/2), 0.0])
    return r, quat_mul(Q0, dq)


def measurement_stream(cfg, n_features, frames, dT=1.0/30.0, seed=1234, noise_seed=1235,
                       sigma_px=None, dtype=np.float32):
    """Returns (pixels0 (N,2), z (frames, N, 2)): the pixels the features are initialised from at
    t = 0 and their noisy observations at t = dT, 2 dT, ..."""
    rng = np.random.default_rng(seed + 7)
    px0 = initial_pixels(cfg, n_features, seed)
    depth = rng.uniform(DEPTH[0], DEPTH[1], size=n_features)
    rays = unproject(cfg, px0)
    R0 = quat2rot(Q0)
    world = (rays * depth[:, None]) @ R0.T
    nrng = np.random.default_rng(noise_seed)
    sigma = float(cfg["sigma_pixel"]) if sigma_px is None else sigma_px
    z = np.empty((frames, n_features, 2))
    for f in range(frames):
        r, q = trajectory((f + 1) * dT)
        pc = (world - r) @ quat2rot(q)            # R^T (y - r)
        z[f] = project(cfg, pc) + nrng.normal(0.0, sigma, size=(n_features, 2))
    return px0, z.astype(dtype)
