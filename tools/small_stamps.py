"""Phase boundaries of k_update_small_onelaunch (ekf_small.hpp) as its workgroup 0 saw them (EKF_SMALL_STAMPS=1; the
100 MHz constant clock, so 10 ns steps), averaged over `steps` updates, and the step time with and without the kernel:
    python tools/small_stamps.py [N] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from __graft_entry__ import load_package
pkg = load_package()
import torch
from ekf_monoslam_amd import synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = pkg.kinect_config()
px0, z = synthetic.measurement_stream(cfg, N, steps + 60, sigma_px=0.5)
d_z = torch.from_numpy(z.reshape(z.shape[0], -1)).cuda().contiguous()
d_idx = torch.arange(N, dtype=torch.int32, device="cuda")
NAMES = ["Sigma + list loads issued", "W = Sigma H^T (row chunks through LDS)", "S = H W + R", "factor (diag_factor_lds)",
         "L^-1 layout (+ workspaces: nu workgroup only)", "y, W rows -> registers", "V = W L^-T", "gate + barrier",
         "V V^T products, mu, q'", "tile epilogue + stores"]


def run(mode, stamps=False):
    os.environ["EKF_SMALL_ONELAUNCH"] = mode
    os.environ["EKF_SMALL_STAMPS"] = "1" if stamps else "0"
    f = pkg.VSlamFilter(cfg, capacity_features=N)
    f.setDt(1 / 30.0)
    for (u, v) in px0:
        f.addFeature((u, v))
    for k in range(50):
        f.predict(); f.update_device(d_z.data_ptr() + k * 8 * N, d_idx.data_ptr(), N, True)
    f.synchronize()
    acc = np.zeros(10)
    if stamps:
        reps = 20
        for rep in range(reps):                        # the LAST update of a run of back-to-back steps (no idle gap in front of it)
            for k in range(steps):
                f.predict(); f.update_device(d_z.data_ptr() + (k % 50) * 8 * N, d_idx.data_ptr(), N, True)
            st = f.peekWorkspace(3, 0, 0, 16, 2).view(np.uint32).astype(np.uint64)
            t = (st[:, 0] | (st[:, 1] << np.uint64(32))).astype(np.float64)
            acc += np.diff(t[:11]) * 0.01
            stage = getattr(run, "stage", np.zeros(4)); run.stage = stage + (t[11:15] - t[1]) * 0.01 / reps
        acc *= steps / reps
        print(f"N = {N}: workgroup 0 of k_update_small_onelaunch, us per phase (the last of {steps} back-to-back steps, mean of 20 runs)")
        for nm, v in zip(NAMES, acc / steps):
            print(f"  {nm:48s} {v:6.2f}")
        print(f"  {'kernel entry -> last store':48s} {acc.sum() / steps:6.2f}")
        print("  chunk c staged at (us after 'loads issued'):", " ".join(f"{v:.2f}" for v in run.stage if 0 < v < 1e3))
    f.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        f.predict(); f.update_device(d_z.data_ptr() + (k % 50) * 8 * N, d_idx.data_ptr(), N, True)
    f.synchronize()
    dt = (time.perf_counter() - t0) / steps
    lc = {k: v for k, v in f.launch_counts().items() if v}
    print(f"EKF_SMALL_ONELAUNCH={mode}{' (stamps on)' if stamps else ''}: {1e6 * dt:.1f} us per predict + update  {lc}")
    f.close()


run("1", True)
run("1")
run("0")
run("1")
run("0")
