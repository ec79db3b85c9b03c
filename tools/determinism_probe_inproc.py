"""The two-rank rig INSIDE ONE PROCESS: two sharded filter handles (rank 0 and rank 1 of a world of
two), each driven by its own host thread, sharing the one GPU through ONE HIP context; the all-gather callback is an
in-process rendezvous that stages the slots through host memory exactly as the gloo rig of the tests does (synchronise the
library's stream, copy the own slot to the host, meet, copy both slots back on the stream).  Every rank repeats the same
short run REPS times on fresh filters and compares mu and its own rows of Sigma after every frame, bit for bit, with its
first repetition -- the run of tools/determinism_probe_sharded.py, which with two PROCESSES on one GPU showed 2-5 deviating
repetitions per 1280 frames on the library of commit 7fa5cff.
    python tools/determinism_probe_inproc.py [N] [REPS] [FRAMES] [RESIZE_EVERY] [package root]
The package root (default: this repository) selects the library: tools/_r5_7fa5cff = the sources of commit 7fa5cff
(git worktree add /tmp/old 7fa5cff; make; copy __graft_entry__.py, include/, the package's *.py and lib/)."""
import ctypes as C
import os, sys, threading, time, traceback
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 80
FRAMES = int(sys.argv[3]) if len(sys.argv) > 3 else 16
EVERY = int(sys.argv[4]) if len(sys.argv) > 4 else 2
ROOT = os.path.abspath(sys.argv[5]) if len(sys.argv) > 5 else os.path.dirname(HERE)
sys.path[:0] = [ROOT]
from __graft_entry__ import load_package
pkg = load_package()
import torch
from ekf_monoslam_amd import sharded, synthetic
WORLD = 2
FULL = int(os.environ.get("PROBE_FULL", "0"))          # 1: compare the own rows of Sigma too (72 MB per frame through the host)
cfg = pkg.kinect_config()
px0, z = synthetic.measurement_stream(cfg, N, FRAMES, sigma_px=0.5)


class InProcGather:
    """ekf_allgather_fn of one rank; the two instances share `meet` (a barrier) and `slots` (host tensors)."""
    meet = threading.Barrier(WORLD)
    slots = [None] * WORLD

    def __init__(self, rank):
        self.rank = rank
        self.error = None
        self._streams = {}
        self.c_callback = sharded.ALLGATHER_FN(self._call)

    def _call(self, ctx, d_send, d_recv, nbytes, stream):
        try:
            st = self._streams.get(stream)
            if st is None:
                st = torch.cuda.ExternalStream(stream, device="cuda:0") if stream else torch.cuda.default_stream(0)
                self._streams[stream] = st
            send = torch.as_tensor(sharded._DevArray(d_send, nbytes), device="cuda:0")
            recv = torch.as_tensor(sharded._DevArray(d_recv, nbytes * WORLD), device="cuda:0")
            with torch.cuda.stream(st):
                InProcGather.slots[self.rank] = send.cpu()           # (synchronises the library's stream, as the gloo rig does)
                InProcGather.meet.wait()
                both = torch.cat(InProcGather.slots)
                recv.copy_(both)
                st.synchronize()
                InProcGather.meet.wait()                             # nobody replaces its slot before the other has read it
            return 0
        except Exception as e:
            self.error = e
            traceback.print_exc()
            try:
                InProcGather.meet.abort()
            except Exception:
                pass
            return 1


results = [None] * WORLD
t0 = time.time()


def worker(rank):
    try:
        torch.cuda.set_device(0)

        def one():
            f = pkg.VSlamFilter(cfg, capacity_features=N + 64, dtype=np.float32)
            f.setDt(1 / 30.0)
            for (u, v) in px0:
                assert f.addFeature((u, v)) == 1
            ag = InProcGather(rank)
            f._allgather = ag
            f._check(f._lib.ekf_shard_configure(f._h, rank, WORLD, ag.c_callback, None))
            rng = np.random.default_rng(1236)
            sid = np.arange(N)
            out = []
            for k in range(FRAMES):
                f.predict()
                pr = f.predictions()
                vis = pr[1]
                sel = np.nonzero(vis.astype(bool) & (sid >= 0))[0].astype(np.int32)
                f.update(z[k][sid[sel]].reshape(-1), sel)
                if (k + 1) % EVERY == 0:
                    n_now = f.numOfFeatures()
                    drop = sorted(rng.choice(n_now, size=n_now // 100, replace=False).tolist())
                    f.removeFeatures(drop)
                    sid = np.delete(sid, drop)
                    for _ in range(len(drop)):
                        assert f.addFeature((float(rng.uniform(20, 300)), float(rng.uniform(20, 220)))) == 1
                    sid = np.concatenate([sid, -np.ones(len(drop), np.int64)])
                n = f.stateDim()
                if FULL:
                    info = sharded.shard_info(f)
                    S = np.vstack([f.getSigmaBlock(0, 0, 14, n),
                                   f.getSigmaBlock(int(info.row_begin), 0, int(info.row_end - info.row_begin), n)])
                else:           # mu and the camera rows: every deviation the two-process rig logged shows in mu
                    S = f.getSigmaBlock(0, 0, 14, n)
                out.append((f.getFullState().copy(), S))
            f.close()
            return out

        ref = one()
        bad = 0
        for rep in range(1, REPS):
            cur = one()
            for k, ((m0, s0), (m1, s1)) in enumerate(zip(ref, cur)):
                dm = np.flatnonzero(m0 != m1)
                ds = np.argwhere(s0 != s1)
                if dm.size or len(ds):
                    bad += 1
                    mag = float(np.max(np.abs(s0.astype(np.float64) - s1))) if len(ds) else 0.0
                    print(f"rank {rank} rep {rep} frame {k}: mu differs in {dm.size} entries (first {dm[:4].tolist()}: {m0[dm[:4]].tolist()} vs "
                          f"{m1[dm[:4]].tolist()}); own rows of Sigma: {len(ds)} entries differ, max |diff| {mag:.3e}", flush=True)
                    break
            if rep % 5 == 0:
                print(f"rank {rank}: {rep + 1} repetitions ({(rep + 1) * FRAMES} frames) done, {bad} deviating, {time.time() - t0:.0f} s", flush=True)
        print(f"rank {rank}/{WORLD} IN ONE PROCESS, N = {N}: {REPS} repetitions of {FRAMES} frames: {bad} differ from the first", flush=True)
        results[rank] = bad
    except Exception:
        traceback.print_exc()
        try:
            InProcGather.meet.abort()
        except Exception:
            pass


t0 = time.time()
ths = [threading.Thread(target=worker, args=(r,)) for r in range(WORLD)]
for t in ths:
    t.start()
for t in ths:
    t.join()
print(f"library {os.path.join(ROOT, 'ekf-monoslam_for_3d-reconstruction_amd', 'lib')}: deviating repetitions per rank {results}, {time.time() - t0:.0f} s", flush=True)
