"""Test-only helper processes: a free-running oracle over a whole measurement stream, run beside the test that
drives the GPU (spawned, never forked: the child imports numpy and the oracle only, nothing of HIP)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def predict_no_St(f, t_ctl=(0, 0, 0), r_ctl=(0, 0, 0), vcontrol=False):
    """DenseFilter.predict without the full St = H Sigma H^T + R it forms at the end (vR.cpp:598): the update
    recomputes St for the measured set anyway (vR.cpp:1268), and at N = 200 the extra product is a third of a frame."""
    import ekf_oracle as o
    Ft, Q = f._motion(t_ctl, r_ctl, vcontrol)
    f.Ft, f.Q = Ft, Q
    f.predict_covariance(Ft, Q)
    f.mu[0:13] = o.predict_state(f.mu[0:13], t_ctl, r_ctl, f.dT, f.T)
    if f.camera_dim == 14:
        f.map_scale = f.mu[13]
    f.measure()


def free_running_oracle(px0, zs, dtype_name, dT, queue):
    """Structured oracle of `dtype_name` over every frame of zs (frames, N, 2), measuring the visible features.
    Puts (mu, Sigma, number of frames in which some feature was not visible) on the queue."""
    import ekf_oracle as o
    from threadpoolctl import threadpool_limits
    threadpool_limits(limits=4)
    T = np.dtype(dtype_name).type
    f = o.StructuredFilter(o.Config.kinect(), T)
    f.dT = dT
    for (u, v) in px0:
        assert f.add_feature(u, v) == 1
    partial = 0
    for k in range(zs.shape[0]):
        predict_no_St(f)
        vis = f.visible_indices()
        partial += int(len(vis) != len(f.features))
        f.update(zs[k][vis].reshape(-1).astype(T), vis)
    queue.put((f.mu, f.Sigma, partial))
