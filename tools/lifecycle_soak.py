"""Create -> add features -> a few predict + update steps -> read back -> destroy, many times in one process (the pattern of
the two GPU tests that hung once each in round 6): python tools/lifecycle_soak.py [cycles].  A stall dumps the
Python stack (faulthandler) after 60 s and exits non-zero."""
import faulthandler, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import synthetic
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = pkg.kinect_config()
streams = {n: synthetic.measurement_stream(cfg, n, 3, sigma_px=0.5) for n in (20, 32, 70, 200, 640)}
t0 = time.perf_counter()
ref = {}
for c in range(cycles):
    n = (20, 32, 70, 200, 640)[c % 5]
    px0, z = streams[n]
    faulthandler.dump_traceback_later(60, exit=True)
    f = pkg.VSlamFilter(cfg, capacity_features=n)
    f.setDt(1 / 30.0)
    for (u, v) in px0:
        f.addFeature((u, v))
    idx = np.arange(n, dtype=np.int32)
    for k in range(3):
        f.predict()
        f.update(z[k].reshape(-1), idx)
    mu = f.getFullState()
    f.close()
    faulthandler.cancel_dump_traceback_later()
    if n in ref:
        assert np.array_equal(ref[n], mu), (c, n)
    ref[n] = mu
    if c % 50 == 49:
        print(f"cycle {c + 1}: {time.perf_counter() - t0:.1f} s, results bit-identical to the first cycle of each size", flush=True)
print("lifecycle soak OK")
