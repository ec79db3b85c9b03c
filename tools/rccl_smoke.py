"""One-command smoke test of the multi-GPU path over RCCL, for the day a node with more than one GPU is at hand:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port 29511 \\
        tools/rccl_smoke.py [--features 200] [--frames 6] [--backend nccl]

(G = 2 .. 8, one rank per GPU; nothing touches a GPU before the rank has picked its device from LOCAL_RANK.)
Every rank builds the same N-feature map, switches it to row-panel sharding over torch.distributed (backend "nccl" =
RCCL over xGMI) and runs `frames` frames of predict + update -- all features measured, then a measured subset with the
plane rows, a removal / addition, a conversion pass, the two-stage update of the reference's update() flow and the map
export (points table with archived patches, per-feature XYZ getters: every rank must return the OWNERS' numbers).  Every rank
ALSO runs the plain single-GPU filter on the same calls and compares its camera rows + own rows of Sigma and the
replicated state: in fp64 (bound 1e-8: the protocol is exact up to the order of sums) and in fp32 (bound 5e-3: the two
paths may cut S into different column chunks, and an fp32 filter amplifies rounding differences of 1e-7 to 1e-4 .. 1e-3
within a few frames, tools/shard_cadence_probe.py).  The maximum error over the ranks is all-reduced; the exit code is
non-zero on a mismatch (or on any exception in any rank).
--backend gloo rehearses the same script with several ranks sharing ONE GPU (collectives through host memory).
The distributed Cholesky chain (round 6) starts at 40 block steps (N >= 2500): `--features 4000`, or EKF_SHARD_DIST_MIN_BLOCKS=2 in
the environment to put this script's small map through it."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--features", type=int, default=200)
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    import torch
    import torch.distributed as dist
    ndev = torch.cuda.device_count()                       # (counting devices does not initialise the GPU)
    dev = local % max(ndev, 1) if args.backend == "nccl" else 0
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"rccl_smoke: {world} ranks over RCCL need {world} GPUs, this node shows {ndev} (use --backend gloo to rehearse)")
    torch.cuda.set_device(dev)
    dist.init_process_group(args.backend, rank=rank, world_size=world)
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import sharded, synthetic
    cfg = pkg.kinect_config()
    N = args.features
    ok_all = True
    for dtype, tol in ((np.float64, 1e-8), (np.float32, 5e-3)):
        ok_all = run(args, pkg, sharded, synthetic, cfg, N, dtype, tol, rank, world, dev, torch, dist) and ok_all
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok_all else 1)


def run(args, pkg, sharded, synthetic, cfg, N, dtype, tol, rank, world, dev, torch, dist):
    px0, z = synthetic.measurement_stream(cfg, N, args.frames, sigma_px=0.5, dtype=dtype)
    rng = np.random.default_rng(1236)

    def build(shard):
        f = pkg.VSlamFilter(cfg, capacity_features=N + 16, dtype=dtype, device=dev)
        f.setDt(1.0 / 30.0)
        for (u, v) in px0:
            assert f.addFeature((u, v)) == 1
        if shard:
            sharded.configure(f, rank, world, dev)
        return f

    plain, shd = build(False), build(True)
    sid = np.arange(N)
    drop = sorted(rng.choice(N, size=max(1, N // 50), replace=False).tolist())
    adds = [(float(rng.uniform(20, 300)), float(rng.uniform(20, 220))) for _ in drop]
    worst = 0.0
    for k in range(args.frames):
        for f in (plain, shd):
            f.predict()
        h, vis, rem, S2 = plain.predictions()
        sel = np.nonzero(vis & (sid >= 0))[0].astype(np.int32)
        if k % 3 == 1 and 'subset' not in os.environ.get('SMOKE_SKIP', ''):
            sel = sel[::2]
        zz = z[k][sid[sel]].reshape(-1)
        for f in (plain, shd):
            if k == args.frames - 1:
                f.updateTwoStage(zz, sel, plane_constraint=False, seed=0)
            else:
                f.update(zz, sel, plane_constraint=(k % 3 == 1 and 'plane' not in os.environ.get('SMOKE_SKIP', '')))
        if k == 1 and 'remove' not in os.environ.get('SMOKE_SKIP', ''):
            for f in (plain, shd):
                f.removeFeatures(drop)
                for (u, v) in adds:
                    assert f.addFeature((u, v)) == 1
            sid = np.concatenate([np.delete(sid, drop), -np.ones(len(drop), np.int64)])
        if k == 2 and 'convert' not in os.environ.get('SMOKE_SKIP', ''):
            a, b = plain.convert2XYZ_ifLinearAll(), shd.convert2XYZ_ifLinearAll()
            assert a == b, (a, b)
        for f in (plain, shd):
            f.synchronize()
        info = sharded.shard_info(shd)
        rows = np.r_[0:14, info.row_begin:info.row_end]
        n = plain.stateDim()
        S_p = np.concatenate([plain.getSigmaBlock(int(r), 0, 1, n) for r in rows]) if len(rows) < 64 else plain.getFullSigma()[rows]
        S_s = np.concatenate([shd.getSigmaBlock(int(r), 0, 1, n) for r in rows]) if len(rows) < 64 else shd.getFullSigma()[rows]
        e_mu = float(np.linalg.norm(shd.getFullState() - plain.getFullState()) / np.linalg.norm(plain.getFullState()))
        e_S = float(np.linalg.norm(S_s - S_p) / max(np.linalg.norm(S_p), 1e-300))
        pad, asym, big = shd.checkInvariants()               # capacity > N: nothing may leak outside the live block,
        worst = max(worst, e_S, 10 * e_mu, float(pad))       # also after a conversion / removal has shrunk n
        print(f"[rccl_smoke rank {rank}/{world} {np.dtype(dtype).name}] frame {k}: features {info.f_begin}..{info.f_end} rows {info.row_begin}..{info.row_end} "
              f"rel|mu| {e_mu:.2e} rel|Sigma rows| {e_S:.2e} rebalances {info.rebalances}", flush=True)
    # the map export under sharding (a feature's covariance block is valid on its owner only: the owners' diagonal blocks
    # are gathered inside these calls): removal of XYZ features that were found often enough to be archived
    # (vslamRansac.cpp:394-404), then the points table of getPointsFeatures and the per-feature getters on EVERY rank
    # against the plain path -- ranks that do not own a feature must return its owner's numbers
    lay = plain.featureLayout()
    xyz = [i for i in range(plain.numOfFeatures()) if lay[1][i] != 0]
    if 'export' not in os.environ.get('SMOKE_SKIP', ''):
        for f in (plain, shd):
            for i in range(plain.numOfFeatures()):
                f.setFeatureMeta(i, n_find=9)
        gone = [xyz[0], xyz[len(xyz) // 2], xyz[-1]] if len(xyz) >= 6 else []
        for f in (plain, shd):
            if gone:
                f.removeFeatures(sorted(set(gone)))
        assert plain.numArchived() == shd.numArchived() == len(set(gone))
        tp, ts = plain.getPointsTable(), shd.getPointsTable()
        pp, ps = plain.getPointsFeatures(True), shd.getPointsFeatures(True)
        last = plain.numOfFeatures() - 1
        fx = [np.concatenate([np.ravel(a) for a in f.featureXYZ(i)]) for f in (plain, shd) for i in (0, last)]
        def rel(a, b):                                       # positions and covariances on their own scales
            return max(float(np.abs(a[..., :3] - b[..., :3]).max()) / max(float(np.abs(b[..., :3]).max()), 1e-300),
                       float(np.abs(a[..., 3:] - b[..., 3:]).max()) / max(float(np.abs(b[..., 3:]).max()), 1e-300))
        e_tab, e_pts = rel(ts, tp), rel(ps, pp)
        e_fx = max(rel(fx[2], fx[0]), rel(fx[3], fx[1]))
        # (the converted covariance of an inverse-depth point, J Sigma_6x6 J^T with J ~ 1 / rho^2, amplifies the fp32
        # differences between two chunk plans by orders of magnitude for far points: bounded in fp64, printed in fp32)
        worst = max(worst, e_tab, e_fx, e_pts if dtype == np.float64 else 0.0)
        print(f"[rccl_smoke rank {rank}/{world} {np.dtype(dtype).name}] map export: {len(xyz)} XYZ features, {len(set(gone))} archived, "
              f"table {tp.shape[0]} rows max|diff| {e_tab:.2e}, points {e_pts:.2e}, featureXYZ {e_fx:.2e}", flush=True)
    t = torch.tensor([worst], dtype=torch.float64, device=(f"cuda:{dev}" if args.backend == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = bool(t.item() < tol) and bool(np.isfinite(t.item()))
    if rank == 0:
        print(f"[rccl_smoke] backend {args.backend} world {world} N {N} {np.dtype(dtype).name}: worst relative error over the "
              f"ranks {t.item():.2e} (bound {tol:.0e}) -> {'OK' if ok else 'MISMATCH'}", flush=True)
    for f in (plain, shd):
        f.close()
    return ok


if __name__ == "__main__":
    main()
