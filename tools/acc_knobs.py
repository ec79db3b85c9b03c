"""Accuracy of the N = 1000 default pipeline against the fp64 oracle for a list of environment settings (the knobs are read
when a filter is created):  python tools/acc_knobs.py "EKF_W_RECOMPUTE=0" "EKF_LAZY_TRAILING=0" ...   ("" = defaults).
Per setting: relative Frobenius error of mu and Sigma after each of FRAMES frames of the bench stream (all features measured)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from __graft_entry__ import load_package
pkg = load_package()
import ekf_oracle as o
import oracle_worker
from ekf_monoslam_amd import synthetic
N = int(os.environ.get("ACC_N", "1000")); FRAMES = int(os.environ.get("ACC_FRAMES", "2"))
px0, z = synthetic.measurement_stream(pkg.kinect_config(), N, FRAMES, sigma_px=0.5, dtype=np.float32)
ref = o.StructuredFilter(o.Config.kinect(), np.float64); ref.dT = 1.0 / 30.0
assert ref.add_features(px0) == N
states = []
for k in range(FRAMES):
    oracle_worker.predict_no_St(ref)
    ref.update(z[k].reshape(-1).astype(np.float64), list(range(N)))
    states.append((ref.mu.copy(), ref.Sigma.copy()))
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
settings = sys.argv[1:] or [""]
keys = {kv.split("=")[0] for st in settings for kv in st.split()}
for st in settings:
    for k in keys: os.environ.pop(k, None)
    for kv in st.split():
        k, v = kv.split("="); os.environ[k] = v
    f = pkg.VSlamFilter(pkg.kinect_config(), capacity_features=N); f.setDt(1.0 / 30.0)
    for (u, v) in px0: f.addFeature((u, v))
    out = []
    for k in range(FRAMES):
        f.predict(); f.update(z[k].reshape(-1), np.arange(N, dtype=np.int32)); f.synchronize()
        out.append("frame %d: mu %.2e Sigma %.2e" % (k, rel(f.getFullState(), states[k][0]), rel(f.getFullSigma(), states[k][1])))
    print("[%-44s] %s" % (st or "defaults", " | ".join(out)), flush=True)
    f.close()
