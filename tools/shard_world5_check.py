import os, sys
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle"), os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests")]
import numpy as np
import torch.multiprocessing as mp
import test_sharded as t
from __graft_entry__ import load_package
if __name__ == "__main__":
    pkg = load_package()
    from ekf_monoslam_amd import synthetic
    n_feat, frames, world = 600, 2, 5
    px0, z = synthetic.measurement_stream(pkg.kinect_config(), n_feat, frames, sigma_px=0.5)
    z_np = np.ascontiguousarray(z.reshape(frames, -1), np.float32)
    mu_p, S_p = t._plain_hip_run(n_feat, frames, z_np, px0)
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(t._gpu_worker, args=(world, t.free_port(), n_feat, frames, z_np, out, px0), nprocs=world, join=True)
    for rank in range(world):
        mu, rows, S_rows = out[rank]
        print(rank, t.relf(mu, mu_p), t.relf(S_rows, S_p[rows]))
