"""The sharded step's determinism: WORLD ranks share the one GPU (collectives over gloo, as in the suite), and every rank
repeats the same short run (N features, every visible surviving feature measured, a removal of 1 % + as many adds every 4
frames) REPS times on fresh filters, hashing mu and its own rows of Sigma after every frame; any difference between
repetitions is printed with the rank and the frame it first shows in.
A deviation is described in full: which rows / columns of the rank's own rows of Sigma differ and by how much, whether the
row was already different BEFORE the step (the resize of the frame before), and the same for mu.
usage: python3 tools/determinism_probe_sharded.py [N] [WORLD] [REPS] [FRAMES] [RESIZE_EVERY]"""
import os, socket, sys, time
import numpy as np
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
WORLD = int(sys.argv[2]) if len(sys.argv) > 2 else 2
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 10
FRAMES = int(sys.argv[4]) if len(sys.argv) > 4 else 12
EVERY = int(sys.argv[5]) if len(sys.argv) > 5 else 4
PEEK = int(os.environ.get("PROBE_PEEK", "0"))
LIGHT = int(os.environ.get("PROBE_LIGHT", "0"))        # compare mu, the camera rows of Sigma and (PROBE_PEEK) W / V only: shorter frames
SLEEP = float(os.environ.get("PROBE_SLEEP", "0"))      # host pause behind every frame (does the timing between frames matter?)


def worker(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import sharded, synthetic
    cfg = pkg.kinect_config()
    px0, z = synthetic.measurement_stream(cfg, N, FRAMES, sigma_px=0.5)

    def one():
        f = pkg.VSlamFilter(cfg, capacity_features=N + 64, dtype=np.float32)
        f.setDt(1 / 30.0)
        for (u, v) in px0:
            assert f.addFeature((u, v)) == 1
        sharded.configure(f, rank, world)
        rng = np.random.default_rng(1236)
        sid = np.arange(N)
        out = []
        last_drop = []
        for k in range(FRAMES):
            frame_drop = []
            f.predict()
            pr = f.predictions(jacobians=bool(PEEK))
            h, vis, rem = pr[0], pr[1], pr[2]
            jac = (pr[4].copy(), pr[5].copy()) if PEEK else None
            sel = np.nonzero(vis.astype(bool) & (sid >= 0))[0].astype(np.int32)
            f.update(z[k][sid[sel]].reshape(-1), sel)
            info = sharded.shard_info(f)
            mp = (2 * len(sel) + 127) // 128 * 128
            wk_r0, wk_r1 = int(info.row_begin), int(info.row_end)
            if (k + 1) % EVERY == 0:
                n_now = f.numOfFeatures()
                drop = sorted(rng.choice(n_now, size=n_now // 100, replace=False).tolist())
                f.removeFeatures(drop)
                last_drop = drop
                frame_drop = drop
                sid = np.delete(sid, drop)
                for _ in range(len(drop)):
                    assert f.addFeature((float(rng.uniform(20, 300)), float(rng.uniform(20, 220)))) == 1
                sid = np.concatenate([sid, -np.ones(len(drop), np.int64)])
            info = sharded.shard_info(f)
            n = f.stateDim()
            if LIGHT:
                rows = np.r_[0:14]
                S = f.getSigmaBlock(0, 0, 14, n)
            else:
                rows = np.r_[0:14, info.row_begin:info.row_end]
                S = np.vstack([f.getSigmaBlock(0, 0, 14, n), f.getSigmaBlock(int(info.row_begin), 0, int(info.row_end - info.row_begin), n)])
            mu_now = f.getFullState().copy()
            if SLEEP > 0:
                time.sleep(SLEEP)
            if PEEK:        # W and V of this frame's update (the resize does not touch them), read AFTER the step's own synchronisation
                wk = [f.peekWorkspace(w, wk_r0, 0, wk_r1 - wk_r0, mp) for w in (0, 1)]
            else:
                wk = [np.zeros((1, 1), np.float32)] * 2
            out.append((mu_now, S, rows, list(last_drop), int(len(sel)), wk, wk_r0, jac, sel.copy(), list(frame_drop)))
        f.close()
        return out

    ref = one()
    bad = 0
    for rep in range(1, REPS):
        cur = one()
        for k, ((m0, s0, r0, dr0, M0, wk0, wr0, jac0, sel0, fd0), (m1, s1, r1, dr1, M1, wk1, wr1, jac1, sel1, fd1)) in enumerate(zip(ref, cur)):
            dm = np.flatnonzero(m0 != m1)
            ds = np.argwhere(s0 != s1)
            if dm.size or len(ds):
                bad += 1
                rr = sorted(set(int(r0[a]) for a in ds[:, 0])) if len(ds) else []
                cc = sorted(set(int(b) for b in ds[:, 1])) if len(ds) else []
                mag = float(np.max(np.abs(s0.astype(np.float64) - s1))) if len(ds) else 0.0
                scale = float(np.max(np.abs(s0)))
                print(f"rank {rank} rep {rep} frame {k} (resize every {EVERY}): mu differs in {dm.size} entries (first {dm[:6].tolist()}: {m0[dm[:6]].tolist()} vs "
                      f"{m1[dm[:6]].tolist()}); own rows of Sigma: {len(ds)} entries differ, {len(rr)} rows (first {rr[:8]}), {len(cc)} columns (first {cc[:8]}), "
                      f"max |diff| {mag:.3e} (max |Sigma| {scale:.3e}); state dim {m0.size}, own rows {int(r0[14])}..{int(r0[-1])}", flush=True)
                for i in dm[:2]:
                    print(f"   mu[{int(i) - 2}..{int(i) + 3}]: first run {m0[i - 2:i + 4].tolist()}, this run {m1[i - 2:i + 4].tolist()}", flush=True)
                for (a, b) in ds[:3]:
                    print(f"   Sigma[{int(r0[a])}, {int(b)}]: {float(s0[a, b])!r} vs {float(s1[a, b])!r}", flush=True)
                # the row with the most differing entries: where along it is the difference largest?
                if len(ds):
                    cnt = np.bincount(ds[:, 0])
                    a = int(np.argmax(cnt))
                    d = np.abs(s0[a].astype(np.float64) - s1[a])
                    top = np.argsort(-d)[:10]
                    print(f"   rank {rank}: row {int(r0[a])} ({cnt[a]} entries differ); largest |diff| at columns "
                          + ", ".join(f"{int(c)} (feature {(int(c) - 14) // 6}.{(int(c) - 14) % 6}: {d[c]:.2e}, ref {float(s0[a, c]):.3e})" for c in top), flush=True)
                    print(f"   rank {rank}: M = {M1}, last removal {dr1}", flush=True)
                for name, a0, a1 in (("W", wk0[0], wk1[0]), ("V", wk0[1], wk1[1])):
                    dd = np.argwhere(a0 != a1)
                    if not len(dd):
                        print(f"   rank {rank}: own rows of {name} of this update (before the resize): identical", flush=True)
                        continue
                    rws = sorted(set(int(x) + wr0 for x in dd[:, 0]))
                    cls = sorted(set(int(x) for x in dd[:, 1]))
                    print(f"   rank {rank}: own rows of {name}: {len(dd)} entries differ, rows {rws[:8]}, columns {cls[0]}..{cls[-1]} ({len(cls)} of them; first {cls[:12]})", flush=True)
                    a = int(dd[0, 0])
                    for c in cls[:6]:
                        print(f"      {name}[{a + wr0}, {c}]: {float(a0[a, c])!r} vs {float(a1[a, c])!r}", flush=True)
                    if name == "W" and jac1 is not None:
                        # the two halves of the sum behind every deviating entry of the FIRST 16 deviating columns: camera part, feature part
                        # (Sigma row from the first run's state after the step; a removal behind the step shifts the indices)
                        i_pre = a + wr0
                        fdrop = np.asarray(sorted(fd0), np.int64)
                        f_i = (i_pre - 14) // 6
                        if f_i in set(fd0):
                            continue
                        i_post = i_pre - 6 * int(np.searchsorted(fdrop, f_i))
                        if LIGHT:           # only the camera rows were kept: Sigma[i, 0:7] = Sigma[0:7, i] (the matrix is exactly symmetric)
                            srow = np.zeros(s0.shape[1], np.float32)
                            srow[:14] = s0[:14, i_post]
                        else:
                            ridx = int(np.flatnonzero(r0 == i_post)[0])
                            srow = s0[ridx].astype(np.float32)
                        same_H = bool(np.array_equal(jac0[0], jac1[0]) and np.array_equal(jac0[1], jac1[1]))
                        print(f"      Jacobians of the two runs identical: {same_H}", flush=True)
                        for c in cls[:16]:
                            kk, comp = c // 2, c % 2
                            feat = int(sel1[kk])
                            if feat in set(fd0):
                                continue
                            pp = 14 + 6 * (feat - int(np.searchsorted(fdrop, feat)))
                            hc = jac1[0][feat, comp].astype(np.float32)
                            hf = jac1[1][feat, comp].astype(np.float32)
                            cam = np.float32(0)
                            for t in range(7):
                                cam = np.float32(np.float64(srow[t]) * np.float64(hc[t]) + np.float64(cam))
                            ft = np.float32(0)
                            for t in range(6):
                                ft = np.float32(np.float64(srow[pp + t]) * np.float64(hf[t]) + np.float64(ft))
                            print(f"      W[{i_pre}, {c}] (slot {kk}, feature {feat}, {'uv'[comp]}): first run {float(a0[a, c]):.4e}, this run {float(a1[a, c]):.4e}; "
                                  f"camera part {float(cam):.4e}, feature part {float(ft):.4e}; S[i,1] {float(srow[1]):.4e} S[i,p+1] {float(srow[pp + 1]):.4e} "
                                  f"Hc[1] {float(hc[1]):.4e} Hf[1] {float(hf[1]):.4e}", flush=True)
                break
    print(f"rank {rank}/{world}, N = {N}: {REPS} repetitions of {FRAMES} frames: {bad} differ from the first", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(worker, args=(WORLD, port), nprocs=WORLD, join=True)
