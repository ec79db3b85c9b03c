import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ekf_oracle as o
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd.sharded import HipShardBackend, ShardedStep
n_feat = 40
cfg = o.Config.kinect()
def mk():
    flt = pkg.VSlamFilter(pkg.kinect_config(), capacity_features=n_feat, dtype=np.float32)
    flt.setDt(1/30)
    for (u, v) in o.synthetic_pixels(cfg, n_feat): flt.addFeature((u, v))
    return flt
a = mk(); b = mk()
a.predict()
h, vis, rem, S2 = a.predictions()
z = (h + 0.3).astype(np.float32).reshape(-1)
Sa = a.innovationCovariance(list(range(n_feat)))
be = HipShardBackend(b, 0, 1)
be.predict()
d_z = torch.from_numpy(z).cuda()
be.innovation(d_z.data_ptr(), n_feat)
b.synchronize()
be.refresh_view()
t = be.tensors()
Sb = t["S"][:, :2*n_feat].cpu().numpy()
print("h diff", np.abs(t["h"].cpu().numpy() - h).max())
print("S rel diff", np.linalg.norm(Sa - Sb) / np.linalg.norm(Sa), "diag min", np.diag(Sb).min(), "view m,m_pad,ldy", be.view.m, be.view.m_pad, be.view.ldy)
print(np.argwhere(np.abs(Sa - Sb) > 1e-3 * np.abs(Sa).max())[:10])
