"""The reference's own per-frame flow at its operating point: ekf_predict -> ekf_get_predictions -> ekf_update_two_stage
(1-point RANSAC, low-innovation update, rescue, second update) with z from the host.
python tools/two_stage_latency.py [N] [frames] [outliers per frame]"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R]
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
cfg = pkg.kinect_config()
px0, z = synthetic.measurement_stream(cfg, N, frames, sigma_px=0.5)
f = pkg.VSlamFilter(cfg, capacity_features=N)
f.setDt(1 / 30.0)
for (u, v) in px0:
    assert f.addFeature((u, v)) == 1
idx = np.arange(N, dtype=np.int32)
zz = z.reshape(frames, -1).astype(np.float32)
OUT = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # outliers per frame (+8 px): they take the rescue path
rng = np.random.default_rng(5)
for k in range(frames):
    for j in rng.choice(N, OUT, replace=False):
        zz[k, 2 * j] += 8.0
tp = tg = tu = 0.0
nli = nhi = 0
for k in range(frames):
    if k == 100:
        f.synchronize(); t0 = time.perf_counter(); tp = tg = tu = 0.0
    a = time.perf_counter(); f.predict()
    b = time.perf_counter(); f.predictions()
    c = time.perf_counter(); li, hi, drawn = f.updateTwoStage(zz[k], idx, seed=1 + k)
    d = time.perf_counter(); tp += b - a; tg += c - b; tu += d - c
    nli += int(np.sum(li)); nhi += int(np.sum(hi))
f.synchronize()
n = frames - 100
dt = (time.perf_counter() - t0) / n
print(f"N={N}: {dt * 1e6:.1f} us per frame ({1 / dt:.0f} frames/s): predict {tp / n * 1e6:.1f}, get_predictions {tg / n * 1e6:.1f}, "
      f"update_two_stage {tu / n * 1e6:.1f} us; low-innovation inliers per frame {nli / frames:.1f}, rescued {nhi / frames:.1f}")
