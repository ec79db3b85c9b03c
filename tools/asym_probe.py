import sys, numpy as np
sys.path.insert(0,'.')
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import synthetic
cfg = pkg.kinect_config()
for N, opt in ((1000, (0, 1)), (1000, (3, 0)), (1000, (3, -1))):
    px0, z = synthetic.measurement_stream(cfg, N, 4, sigma_px=0.5)
    f = pkg.VSlamFilter(cfg, capacity_features=N)
    f.setDt(1/30.)
    for (u, v) in px0: f.addFeature((u, v))
    f.set_option(*opt)
    for k in range(3):
        f.predict()
        a = f.checkInvariants()
        f.update(z[k].reshape(-1), np.arange(N, dtype=np.int32))
        print(N, opt, k, "after predict", a, "after update", f.checkInvariants())
        S = f.getFullSigma()
        d = np.argwhere(S != S.T)
        if len(d):
            print("  asym entries", len(d), "rows", d[:,0].min(), d[:,0].max(), "cols", d[:,1].min(), d[:,1].max(), d[:6].tolist())
