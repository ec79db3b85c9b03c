"""Where does a sharded run leave the plain path?  The resize cadence of tests/test_sharded.py (remove 1 %, add 1 % every
`every` frames) on 2 ranks sharing the GPU, for several run lengths: relative error of every rank's rows, by row group."""
import os, sys
import numpy as np
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_sharded as ts


def worker(rank, world, port, n_feat, frames, every, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import sharded
    flt = pkg.VSlamFilter(pkg.kinect_config(), capacity_features=n_feat + 64, dtype=np.float32)
    flt.setDt(1.0 / 30.0)
    run = ts.n4000_cadence_run(pkg, flt, frames=frames, every=every, n_feat=n_feat)
    next(run); sharded.configure(flt, rank, world); next(run)
    info = sharded.shard_info(flt)
    S = flt.getFullSigma()
    out[rank] = (flt.getFullState(), info.row_begin, info.row_end, S[np.r_[0:14, info.row_begin:info.row_end]], flt.checkInvariants())
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    n_feat = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    from __graft_entry__ import load_package
    pkg = load_package()
    for frames in [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else "5,10,11,20,21,31".split(","))]:
        flt = pkg.VSlamFilter(pkg.kinect_config(), capacity_features=n_feat + 64, dtype=np.float32)
        flt.setDt(1.0 / 30.0)
        run = ts.n4000_cadence_run(pkg, flt, frames=frames, every=every, n_feat=n_feat)
        next(run)
        if os.environ.get("PROBE_REF_SHARD1"):
            from ekf_monoslam_amd import sharded
            sharded.configure(flt, 0, 1)
        next(run)
        mu_p, S_p = flt.getFullState(), flt.getFullSigma()
        out = mp.Manager().dict()
        mp.spawn(worker, args=(2, ts.free_port(), n_feat, frames, every, out), nprocs=2, join=True)
        for rank in range(2):
            mu, r0, r1, S, inv = out[rank]
            ref = S_p[np.r_[0:14, r0:r1]]
            rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
            rows_err = np.linalg.norm(S - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-300)
            worst = np.argsort(rows_err)[-3:]
            print(f"frames {frames:3d} rank {rank} rows [{r0},{r1}) mu {rel(mu, mu_p):.2e} cam rows {rel(S[:14], ref[:14]):.2e} "
                  f"own rows {rel(S[14:], ref[14:]):.2e} own x cam cols {rel(S[14:, :14], ref[14:, :14]):.2e} own x own {rel(S[14:, r0:r1], ref[14:, r0:r1]):.2e} "
                  f"own x lower cols {rel(S[14:, 14:r0], ref[14:, 14:r0]) if r0 > 14 else 0:.2e} own x higher cols {rel(S[14:, r1:], ref[14:, r1:]) if r1 < S.shape[1] else 0:.2e} "
                  f"worst rows {[(int(w), float(rows_err[w])) for w in worst]} invariants {inv}", flush=True)
