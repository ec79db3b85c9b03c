"""Is the step deterministic to the bit?  The same short run (N features, predict / update with every visible surviving feature,
a removal of 1 % + as many adds every 4 frames) repeated REPS times in one process on fresh filters; mu and a sample of rows
of Sigma are hashed after every frame.  Any difference between repetitions is printed with the frame it first shows in.
usage: python3 tools/determinism_probe.py [N] [REPS] [FRAMES]      (environment knobs apply, e.g. EKF_SPLIT_BF16=0)"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
FRAMES = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cfg = pkg.kinect_config()
px0, z = synthetic.measurement_stream(cfg, N, FRAMES, sigma_px=0.5)


def one():
    f = pkg.VSlamFilter(cfg, capacity_features=N + 64, dtype=np.float32)
    f.setDt(1 / 30.0)
    for (u, v) in px0:
        assert f.addFeature((u, v)) == 1
    rng = np.random.default_rng(1236)
    sid = np.arange(N)
    out = []
    for k in range(FRAMES):
        f.predict()
        h, vis, rem, _ = f.predictions()
        sel = np.nonzero(vis.astype(bool) & (sid >= 0))[0].astype(np.int32)
        f.update(z[k][sid[sel]].reshape(-1), sel)
        if (k + 1) % 4 == 0:
            n_now = f.numOfFeatures()
            drop = sorted(rng.choice(n_now, size=n_now // 100, replace=False).tolist())
            f.removeFeatures(drop)
            sid = np.delete(sid, drop)
            for _ in range(len(drop)):
                assert f.addFeature((float(rng.uniform(20, 300)), float(rng.uniform(20, 220)))) == 1
            sid = np.concatenate([sid, -np.ones(len(drop), np.int64)])
        mu = f.getFullState()
        n = f.stateDim()
        rows = np.r_[0:14, n // 2:n // 2 + 64, n - 256:n]
        S = np.vstack([f.getSigmaBlock(int(r), 0, 1, n) for r in rows])
        out.append((mu.copy(), S))
    f.close()
    return out


ref = one()
bad = 0
for rep in range(1, REPS):
    cur = one()
    for k, ((m0, s0), (m1, s1)) in enumerate(zip(ref, cur)):
        dm = np.flatnonzero(m0 != m1)
        ds = np.argwhere(s0 != s1)
        if dm.size or len(ds):
            bad += 1
            print(f"rep {rep} frame {k}: mu differs in {dm.size} entries (first {dm[:4].tolist()}: {m0[dm[:4]].tolist()} vs {m1[dm[:4]].tolist()}), "
                  f"Sigma sample in {len(ds)} entries (first {ds[:3].tolist()})", flush=True)
            break
print(f"N = {N}: {REPS} repetitions of {FRAMES} frames: {bad} repetition(s) differ from the first", flush=True)
