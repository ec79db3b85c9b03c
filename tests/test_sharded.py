"""The N > 1 path: row-panel sharding with all-gathers of H, S, V (+ linearity flags, + Sigma panels on re-balance).

CPU (gloo, world_size 2 and 4): `shard_protocol.ShardProtocol` -- the protocol the library implements, in numpy -- over an
oracle-backed rank that poisons everything it does not own: uneven partitions, measured subsets, XYZ features, the
plane rows, add / remove / convert under sharding and the re-balance must all equal the unsharded oracle.
GPU: the library's sharded step (csrc/ekf_capi.hip) with world 1 in-process, and with 2 / 4 ranks sharing the one GPU
of the box (collectives through gloo and host memory) against the plain HIP path and the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import ekf_oracle as o  # noqa: E402
from helpers import bound, exact_or_anchor_glitch, relf  # noqa: E402


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# ---------------------------------------------------------------------------------------------
# the scenario both sides run: frames of predict + update on changing measured subsets, with XYZ conversions,
# removals, additions (-> imbalance -> re-balance) on the way
# ---------------------------------------------------------------------------------------------
def scenario_events(n_feat, frames, seed=1236):
    """Deterministic schedule: per frame (measured subset rule, plane?, resize event)."""
    rng = np.random.default_rng(seed)
    ev = []
    for k in range(frames):
        e = {"subset": k % 3, "plane": k == 2, "remove": [], "add": [], "convert": k in (1, 4)}
        if k in (2, 5):
            e["remove_frac"] = 0.15
            e["add"] = [(float(rng.uniform(20, 300)), float(rng.uniform(20, 220))) for _ in range(max(3, n_feat // 2))]
        ev.append(e)
    return ev


def pick(vis, rule):
    if rule == 0:
        return list(vis)
    if rule == 1:
        return list(vis[::2])
    return list(vis[1::3]) + ([] if len(vis) < 4 else [])


class PlainOracle:
    """The unsharded oracle driven through the same interface as ShardProtocol."""

    def __init__(self, f):
        self.f = f

    def predict(self):
        f = self.f
        Ft, Q = f._motion((0, 0, 0), (0, 0, 0), False)
        f.predict_covariance(Ft, Q)
        f.mu[0:13] = o.predict_state(f.mu[0:13], (0, 0, 0), (0, 0, 0), f.dT, f.T)
        o.DenseFilter.measure(f)

    def visible(self):
        return self.f.visible_indices()

    def update(self, z, idx, plane):
        self.f.update(z, idx, plane=plane)

    def convert_all(self):
        return self.f.convert2xyz_if_linear_all()

    def remove_features(self, idx):
        for i in reversed(sorted(idx)):
            self.f.remove_feature(i)

    def add_feature(self, u, v):
        return self.f.add_feature(u, v)


def run_scenario(driver, visible, n_feat, frames, seed_z=4000, force_linear=None):
    """Drives `driver` (PlainOracle or ShardProtocol) through the schedule.  `force_linear(frame)` lets the caller
    shrink a few Sigma(rho, rho) so that conversions actually happen."""
    events = scenario_events(n_feat, frames)
    log = []
    for k, e in enumerate(events):
        driver.predict()
        vis = visible()
        idx = pick(vis, e["subset"])
        zrng = np.random.default_rng(seed_z + k)
        z = zrng.normal(0.0, 0.5, size=2 * len(idx))          # noise; the caller adds h
        log.append((k, idx, z, e))
        yield k, idx, z, e


def cpu_scenario(make_driver, n_feat, frames, world=1, rank=0):
    f = o.build_scenario(o.StructuredFilter, o.Config.kinect(), n_feat, np.float64)
    drv, fobj = make_driver(f)
    rng = np.random.default_rng(77)
    events = scenario_events(n_feat, frames)
    for k, e in enumerate(events):
        drv.predict()
        vis = fobj.visible_indices()
        idx = pick(vis, e["subset"])
        hz = np.concatenate([fobj.features[i].h for i in idx]) if idx else np.zeros(0)
        z = hz + np.random.default_rng(4000 + k).normal(0.0, 0.5, size=hz.shape)
        drv.update(z, idx, e["plane"])
        if e["convert"]:
            # make a few inverse-depth features pass the linearity test: the owner's Sigma(rho, rho) decides, so
            # the tweak is applied through mu (replicated): move rho so that the index drops (same on every rank)
            for i in range(1, len(fobj.features), 4):
                ft = fobj.features[i]
                if ft.coding == o.INV:
                    fobj.mu[ft.position_in_state + 5] = 3.0
            drv.convert_all()
        if "remove_frac" in e:
            N = len(fobj.features)
            drop = sorted(rng.choice(N, size=max(1, int(N * e["remove_frac"])), replace=False).tolist())
            drv.remove_features(drop)
            for (u, v) in e["add"]:
                assert drv.add_feature(u, v) == 1
    return fobj


def _cpu_worker(rank, world, port, n_feat, frames, out, dist_chain=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    load_package()
    from shard_protocol import ShardProtocol
    from sharded_common import OracleShardBackend
    holder = {}

    def make(f):
        b = OracleShardBackend(f)
        p = ShardProtocol(b, rank, world, dist_chain=dist_chain)
        holder["p"] = p
        return p, f
    f = cpu_scenario(make, n_feat, frames, world, rank)
    p = holder["p"]
    rows = np.r_[0:f.camera_dim, p.own_rows().start:p.own_rows().stop]
    out[rank] = (f.mu.copy(), rows, f.Sigma[rows].copy(), p.rebalances, list(p.fb),
                 [ft.coding for ft in f.features])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_feat,dist_chain", [(2, 9, False), (4, 14, False), (4, 3, False), (8, 24, False), (8, 5, False),
                                                      (2, 9, True), (4, 14, True), (8, 24, True), (8, 5, True)])
def test_shard_protocol_matches_unsharded_oracle_gloo(world, n_feat, dist_chain):
    """Uneven partitions (9 over 2, 14 over 4, 3 features over 4 ranks: an empty rank; world 8 = the node BASELINE
    configs[3] / [4] name: 24 features, and 5 features over 8 ranks: three empty ranks), subsets, plane rows, XYZ
    conversions, removals and additions, re-balance: every rank's rows must equal the unsharded oracle.
    dist_chain (round 6): the factorisation of S distributed over the ranks -- cyclic row blocks, one all-gather of the panel
    per block step, every rank reading only what it owns or was handed (the rest is NaN in the model) -- with a block of
    2 rows, so that the lists of this test run 5-25 block steps and ranks regularly own nothing of a step."""
    frames = 7
    ref = cpu_scenario(lambda f: (PlainOracle(f), f), n_feat, frames)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_cpu_worker, args=(world, free_port(), n_feat, frames, out, dist_chain), nprocs=world, join=True)
    seen = np.zeros(ref.n, bool)
    for rank in range(world):
        mu, rows, S_rows, rebal, fb, coding = out[rank]
        assert coding == [ft.coding for ft in ref.features]
        assert relf(mu, ref.mu) < 1e-9, rank
        assert np.all(np.isfinite(S_rows)) and relf(S_rows, ref.Sigma[rows]) < 1e-8, rank
        assert fb == out[0][4]                               # every rank tracks the same boundaries
        seen[rows] = True
    assert seen.all()                                        # the panels cover every row of Sigma
    if n_feat >= 9:
        assert any(c == o.XYZ for c in out[0][5])            # the schedule did convert something
        assert out[0][3] >= 1                                # ... and growth at the tail forced a re-balance


def test_partition_balances_rows_not_features():
    from __graft_entry__ import load_package
    load_package()
    from shard_protocol import partition_by_rows
    # 4 XYZ features (3 rows) then 4 inverse-depth ones (6 rows): half of the ROWS is after feature 5
    sizes = [3, 3, 3, 3, 6, 6, 6, 6]
    pos = list(14 + np.concatenate([[0], np.cumsum(sizes)[:-1]]))
    n = 14 + sum(sizes)
    assert partition_by_rows(pos, n, 14, 2) == [0, 5, 8]
    assert partition_by_rows(pos, n, 14, 1) == [0, 8]
    assert partition_by_rows([], 14, 14, 3) == [0, 0, 0, 0]
    fb = partition_by_rows(pos, n, 14, 4)
    assert fb[0] == 0 and fb[-1] == 8 and all(a <= b for a, b in zip(fb, fb[1:]))


# ---------------------------------------------------------------------------------------------
# GPU: the library's sharded step
# ---------------------------------------------------------------------------------------------
def _mk_hip(pkg, n_feat, px0=None, capacity=None, dtype=np.float32):
    """HIP filter on the scenario of o.build_scenario (velocities set, features added in order), or -- with
    px0 -- on the bench's synthetic stream (camera at rest, its pixels)."""
    cfg = o.Config.kinect()
    flt = pkg.VSlamFilter(pkg.kinect_config(), capacity_features=capacity or n_feat, dtype=dtype)
    flt.setDt(1.0 / 30.0)
    if px0 is None:
        mu = flt.getFullState()
        mu[7:10] = (0.3, 0.0, 0.0)
        mu[10:13] = (0.0, 0.05, 0.0)
        flt.setFullState(mu)
        px0 = o.synthetic_pixels(cfg, n_feat)
    for (u, v) in px0:
        assert flt.addFeature((u, v)) == 1
    return flt


def hip_scenario(pkg, flt, ref, frames, dtype):
    """The schedule of cpu_scenario on a HIP filter (sharded or plain) with the oracle `ref` run beside it: the
    measured sets and resize decisions come from the oracle, so every rank (and the plain path) sees the same calls.
    Returns the per-frame measured sets for inspection."""
    rng = np.random.default_rng(77)
    events = scenario_events(len(ref.features), frames)
    for k, e in enumerate(events):
        ref.predict()
        flt.predict()
        vis = ref.visible_indices()
        idx = pick(vis, e["subset"])
        hz = np.concatenate([ref.features[i].h for i in idx]) if idx else np.zeros(0)
        z = (hz + np.random.default_rng(4000 + k).normal(0.0, 0.5, size=hz.shape)).astype(dtype)
        ref.update(z, idx, plane=e["plane"])
        flt.update(z, idx, plane_constraint=e["plane"])
        if e["convert"]:
            for i in range(1, len(ref.features), 4):
                ft = ref.features[i]
                if ft.coding == o.INV:
                    ref.mu[ft.position_in_state + 5] = 3.0
                    flt.setStateSegment(ft.position_in_state + 5, [3.0])
            a = ref.convert2xyz_if_linear_all()
            b = flt.convert2XYZ_ifLinearAll()
            assert a == b, (k, a, b)
        if "remove_frac" in e:
            N = len(ref.features)
            drop = sorted(rng.choice(N, size=max(1, int(N * e["remove_frac"])), replace=False).tolist())
            for i in reversed(drop):
                ref.remove_feature(i)
            flt.removeFeatures(drop)
            for (u, v) in e["add"]:
                assert ref.add_feature(u, v) == 1 and flt.addFeature((u, v)) == 1
        assert flt.numOfFeatures() == len(ref.features) and flt.stateDim() == ref.n


def _gpu_worker(rank, world, port, n_feat, frames, dtype_name, out, capacity=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import sharded
    dtype = np.dtype(dtype_name).type
    flt = _mk_hip(pkg, n_feat, capacity=capacity, dtype=dtype)
    ag = sharded.configure(flt, rank, world)
    ref = o.build_scenario(o.StructuredFilter, o.Config.kinect(), n_feat, dtype)
    hip_scenario(pkg, flt, ref, frames, dtype)
    flt.synchronize()
    info = sharded.shard_info(flt)
    rows = np.r_[0:14, info.row_begin:info.row_end]
    S = flt.getFullSigma()
    pad, asym, big = flt.checkInvariants()
    out[rank] = (flt.getFullState(), rows, S[rows], info.rebalances, (info.f_begin, info.f_end), ag.calls, pad,
                 ref.mu.copy(), ref.Sigma[rows].copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_hip_shard_world1_matches_plain_path_and_oracle(dtype):
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import sharded
    n_feat, frames = 40, 7
    outs = []
    for shard in (False, True):
        flt = _mk_hip(pkg, n_feat, capacity=2 * n_feat + 8, dtype=dtype)
        if shard:
            sharded.configure(flt, 0, 1)
        else:
            flt.set_option(6, 0)       # EKF_OPT_FUSED_LAUNCHES off: the launch-per-kernel sums the sharded path uses too
        ref = o.build_scenario(o.StructuredFilter, o.Config.kinect(), n_feat, dtype)
        hip_scenario(pkg, flt, ref, frames, dtype)
        flt.synchronize()
        outs.append((flt.getFullState(), flt.getFullSigma(), ref))
    (mu_p, S_p, ref), (mu_s, S_s, _) = outs
    f32 = dtype == np.float32
    assert bound("sharded(world 1) mu vs plain path", relf(mu_s, mu_p), 1e-5 if f32 else 1e-12)
    assert bound("sharded(world 1) Sigma vs plain path", relf(S_s, S_p), 2e-4 if f32 else 1e-10)
    assert bound("sharded(world 1) mu vs oracle", relf(mu_s, ref.mu), 5e-5 if f32 else 1e-11)
    assert bound("sharded(world 1) Sigma vs oracle", relf(S_s, ref.Sigma), 2e-3 if f32 else 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("world,n_feat,dtype", [(2, 45, np.float64), (3, 50, np.float32), (4, 60, np.float32)])
def test_hip_shard_ranks_on_one_gpu_dynamic_stream_vs_oracle(world, n_feat, dtype):
    """configs[4] semantics at oracle size: several ranks (sharing the one GPU; collectives through gloo), features
    that do NOT divide by the world size, measured subsets, the plane rows, XYZ conversions, removals and additions
    every few frames, capacity > N, an automatic re-balance -- every rank's rows against the oracle."""
    frames = 7
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gpu_worker, args=(world, free_port(), n_feat, frames, np.dtype(dtype).name, out, 2 * n_feat + 8),
             nprocs=world, join=True)
    f32 = dtype == np.float32
    covered = []
    for rank in range(world):
        mu, rows, S_rows, rebal, frange, calls, pad, mu_ref, S_ref = out[rank]
        assert np.all(np.isfinite(mu)) and np.all(np.isfinite(S_rows))
        assert bound("rank mu vs oracle", relf(mu, mu_ref), 5e-5 if f32 else 1e-11)
        assert bound("rank Sigma rows vs oracle", relf(S_rows, S_ref), 2e-3 if f32 else 1e-9)
        assert pad == 0.0                                    # capacity > N: nothing leaks outside the live block
        assert calls > 0 and rebal >= 1
        covered.append(frange)
    assert covered[0][0] == 0 and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))     # contiguous, complete


def _gpu_static_worker(rank, world, port, n_feat, frames, z_np, px0, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import sharded
    flt = _mk_hip(pkg, n_feat, px0)
    sharded.configure(flt, rank, world)
    d_z = torch.from_numpy(z_np).cuda()
    idx = np.arange(n_feat, dtype=np.int32)
    for k in range(frames):
        flt.predict()
        sharded.shard_update(flt, d_z[k].data_ptr(), idx)
    flt.synchronize()
    info = sharded.shard_info(flt)
    rows = np.r_[0:14, info.row_begin:info.row_end]
    out[rank] = (flt.getFullState(), rows, flt.getFullSigma()[rows])
    dist.barrier()
    dist.destroy_process_group()


def _plain_hip_run(n_feat, frames, z_np, px0=None):
    from __graft_entry__ import load_package
    pkg = load_package()
    flt = _mk_hip(pkg, n_feat, px0)
    idx = list(range(n_feat))
    for k in range(frames):
        flt.predict()
        flt.update(z_np[k], idx)
    return flt.getFullState(), flt.getFullSigma()


@pytest.mark.gpu
@pytest.mark.parametrize("world,n_feat", [(4, 1000), (3, 601)])
def test_hip_shard_full_size_matches_plain_path(world, n_feat):
    """BASELINE configs[3] at full size (N = 1000, n = 6014, 16 block steps in 3 chunks) on 4 ranks sharing the GPU,
    and a size whose panels sit on no tile boundary with an uneven split (601 over 3): every rank's rows must match
    the plain pipelined single-GPU path."""
    frames = 2
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import synthetic
    px0, z = synthetic.measurement_stream(pkg.kinect_config(), n_feat, frames, sigma_px=0.5)
    z_np = np.ascontiguousarray(z.reshape(frames, -1), np.float32)
    mu_p, S_p = _plain_hip_run(n_feat, frames, z_np, px0)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gpu_static_worker, args=(world, free_port(), n_feat, frames, z_np, px0, out), nprocs=world, join=True)
    seen = np.zeros(14 + 6 * n_feat, bool)
    for rank in range(world):
        mu, rows, S_rows = out[rank]
        assert np.all(np.isfinite(mu))
        assert bound("rank mu vs plain path", relf(mu, mu_p), 2e-5)
        assert bound("rank Sigma rows vs plain path", relf(S_rows, S_p[rows]), 5e-4)
        seen[rows] = True
    assert seen.all()


def _gpu_env_worker(rank, world, port, n_feat, frames, z_np, px0, env, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.update(env)                                   # tuning knobs are read when the filter is created
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import sharded
    flt = _mk_hip(pkg, n_feat, px0)
    sharded.configure(flt, rank, world)
    d_z = torch.from_numpy(z_np).cuda()
    idx = np.arange(n_feat, dtype=np.int32)
    for k in range(frames):
        flt.predict()
        sharded.shard_update(flt, d_z[k].data_ptr(), idx)
    flt.synchronize()
    info = sharded.shard_info(flt)
    rows = np.r_[0:14, info.row_begin:info.row_end]
    out[rank] = (flt.getFullState(), rows, flt.getFullSigma()[rows], flt.getGain()[rows], flt.launch_counts())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,n_feat", [(2, 330), (3, 601), (5, 400), (2, 1000)])
def test_hip_shard_distributed_chain_is_bit_identical(world, n_feat):
    """Round 6: from 32 block steps on the factorisation of S is DISTRIBUTED -- a rank keeps only its
    own 128-row blocks of the trailing matrix (cyclic), every diagonal block and the inverse strip up to date, and one
    all-gather per block step hands the panel round (`Filter::dist_chain_steps`).  Every tile is the replicated chain's
    tile, so mu, the rank's rows of Sigma and the gain (which needs ALL of L on every rank) equal the replicated chain's
    to the last bit.  Forced on here at 6-10 block steps (EKF_SHARD_DIST_MIN_BLOCKS=2), several chunks, uneven ownership
    (block counts not divisible by the world size), ranks that own no block of the last steps; at N = 1000 (16 steps) the
    early steps' updates are large enough to take the fused launch (k_trail_diag: the rank's blocks + the next factor)."""
    frames = 2
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import synthetic
    px0, z = synthetic.measurement_stream(pkg.kinect_config(), n_feat, frames, sigma_px=0.5)
    z_np = np.ascontiguousarray(z.reshape(frames, -1), np.float32)
    runs = []
    for env in ({"EKF_SHARD_DIST_CHAIN": "0"}, {"EKF_SHARD_DIST_CHAIN": "1", "EKF_SHARD_DIST_MIN_BLOCKS": "2"}):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_gpu_env_worker, args=(world, free_port(), n_feat, frames, z_np, px0, env, out), nprocs=world, join=True)
        runs.append({r: out[r] for r in range(world)})
    nblk = (2 * n_feat + 127) // 128
    for rank in range(world):
        mu0, rows0, S0, K0, c0 = runs[0][rank]
        mu1, rows1, S1, K1, c1 = runs[1][rank]
        assert c0["chain_dist_gather"] == 0 and c1["chain_dist_gather"] == frames * (nblk - 1), (c0, c1)
        assert n_feat < 1000 or c1["chain_trail_diag"] > 0, c1
        assert np.array_equal(rows0, rows1)
        assert np.all(np.isfinite(mu1)) and np.all(np.isfinite(S1))
        assert np.array_equal(mu0, mu1), float(np.max(np.abs(mu0 - mu1)))
        assert np.array_equal(S0, S1), float(np.max(np.abs(S0 - S1)))
        assert np.array_equal(K0, K1), float(np.max(np.abs(K0 - K1)))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 5])
def test_hip_shard_n1000_ranks_match_fp64_oracle(world):
    """BASELINE configs[3] against the ORACLE, not only against the plain HIP path (VERDICT r2 next #1b): N = M = 1000,
    fp32, two frames of the bench stream on 2 and on 5 ranks sharing the GPU (the box allows six GPU processes at a
    time and this pytest process is one of them, so 8 ranks cannot run here) -- every rank's camera rows + own rows of
    Sigma and its replicated mu against the fp64 structured oracle, at the ceilings of the single-GPU fp32 tests."""
    from helpers import n1000_oracle
    n_feat, frames = 1000, 2
    px0, z, states = n1000_oracle(frames)
    mu_ref, S_ref = states[frames]
    z_np = np.ascontiguousarray(z.reshape(frames, -1), np.float32)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gpu_static_worker, args=(world, free_port(), n_feat, frames, z_np, px0, out), nprocs=world, join=True)
    seen = np.zeros(14 + 6 * n_feat, bool)
    for rank in range(world):
        mu, rows, S_rows = out[rank]
        assert np.all(np.isfinite(mu)) and np.all(np.isfinite(S_rows))
        # the ceilings of test_n1000_default_pipeline_matches_fp64_oracle for the second frame
        assert bound("rank mu vs fp64 oracle", relf(mu, mu_ref), 1e-5)
        assert bound("rank Sigma rows vs fp64 oracle", relf(S_rows, S_ref[rows]), 1e-4)
        own = rows[14:]
        assert bound("rank own Sigma rows, feature columns, vs fp64 oracle", relf(S_rows[14:, 14:], S_ref[own][:, 14:]), 1e-4)
        seen[rows] = True
    assert seen.all()


def _gpu_flow_worker(rank, world, port, n_feat, dtype_name, out):
    """The reference's whole update() flow on a SHARDED filter: every piece against the oracle."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import sharded
    dtype = np.dtype(dtype_name).type
    flt = _mk_hip(pkg, n_feat, capacity=n_feat + 8, dtype=dtype)
    sharded.configure(flt, rank, world)
    ref = o.build_scenario(o.StructuredFilter, o.Config.kinect(), n_feat, dtype)
    # fp32 runs: the fp64 oracle beside the fp32 one, on the same measurements and the same inlier / rescue decisions --
    # the truth the fp32 answers are measured against (VERDICT r3 next #1c: is 4.8e-5 on mu the HIP path or the oracle?)
    ref64 = o.build_scenario(o.StructuredFilter, o.Config.kinect(), n_feat, np.float64) if dtype == np.float32 else None
    res = {}
    # frame 1: the pieces one by one
    ref.predict()
    flt.predict()
    if ref64 is not None:
        ref64.predict()
    vis = ref.visible_indices()
    z = o.synthetic_measurements(ref, vis, sigma=1.0).reshape(-1, 2)
    z[5] += 7.0                                          # outside 2 sigma_px of the prediction, inside a wide chi2 gate
    z[9] += 60.0                                         # a gross mismatch
    h, visg, rem, S2 = flt.predictions()                 # the 2x2 St blocks need every owner's rows: gathered
    S2_ref = np.stack([ref.innovation_covariance([i]) for i in range(n_feat)])
    res["S2"] = relf(S2, S2_ref)
    res["ellipses"] = bool(np.array_equal(flt.searchEllipses(), np.array([o.ellipse_parameters(S2_ref[i], ref.cfg.sigma_size)
                                                                           for i in range(n_feat)])))
    St = flt.innovationCovariance(vis[::2], plane_constraint=True)
    res["St"] = relf(St, ref.innovation_covariance(vis[::2], plane=True))
    counts, best, inl = flt.ransac1Point(z, vis)
    counts_ref, mask_ref = o.ransac_1point(ref, z, vis)
    res["counts"] = bool(np.array_equal(counts, counts_ref)) and int(best) == int(np.argmax(counts_ref))
    res["inliers"] = int((inl != mask_ref[best]).sum())
    mu_before = ref.mu.copy()
    cam_before = flt.getState()[:7]
    li = [vis[k] for k in range(len(vis)) if inl[k]]
    ref.update(z[inl].reshape(-1), li)
    flt.update(z[inl].reshape(-1), li)
    if ref64 is not None:
        assert ref64.visible_indices() == vis
        mu_before64 = ref64.mu.copy()
        ref64.update(z[inl].reshape(-1).astype(np.float64), li)
    rest = [vis[k] for k in range(len(vis)) if not inl[k]]
    hi_ref, chi2 = o.rescue_high_innovation(ref, mu_before, z[~inl], rest, return_chi2=True)
    thr = float(np.sqrt(chi2[rest.index(vis[5])] * chi2[rest.index(vis[9])]))
    hi_ref = o.rescue_high_innovation(ref, mu_before, z[~inl], rest, threshold=thr)
    hi = flt.rescueHighInnovation(cam_before, z[~inl], rest, thr)
    res["rescue"] = list(map(bool, hi)) == list(map(bool, hi_ref)) and bool(hi[rest.index(vis[5])]) and not bool(hi[rest.index(vis[9])])
    sel = [rest[k] for k in range(len(rest)) if hi[k]]
    zz = z[~inl][np.asarray(hi, bool)].reshape(-1)
    ref.update(zz, sel)
    if ref64 is not None:
        hi64 = o.rescue_high_innovation(ref64, mu_before64, z[~inl].astype(np.float64), rest, threshold=thr)
        res["decisions64"] = list(map(bool, hi64)) == list(map(bool, hi_ref))
        ref64.update(zz.astype(np.float64), sel)
    # (device-resident list under sharding: the same update through ekf_update_device)
    d_z = torch.from_numpy(np.ascontiguousarray(zz, dtype)).cuda()
    d_i = torch.from_numpy(np.asarray(sel, np.int32)).cuda()
    flt.update_device(d_z.data_ptr(), d_i.data_ptr(), len(sel))
    # frames 2-3: the composite call, with the reference's draw loop (seed) and the plane rows
    for k in range(2):
        ref.predict()
        flt.predict()
        vis = ref.visible_indices()
        z = o.synthetic_measurements(ref, vis, seed=2000 + k, sigma=0.7).reshape(-1, 2)
        z[3 + k] += 9.0
        li_r, hi_r, drawn_r = o.update_two_stage(ref, z, vis, plane=(k == 1), seed=(0 if k == 0 else 77))
        li_g, hi_g, drawn_g = flt.updateTwoStage(z, vis, plane_constraint=(k == 1), seed=(0 if k == 0 else 77))
        res[f"two_stage_{k}"] = bool(np.array_equal(li_r, li_g) and np.array_equal(hi_r, hi_g) and drawn_r == drawn_g)
        if ref64 is not None:
            ref64.predict()
            li_d, hi_d, drawn_d = o.update_two_stage(ref64, z.astype(np.float64), vis, plane=(k == 1), seed=(0 if k == 0 else 77))
            res["decisions64"] = bool(res["decisions64"] and np.array_equal(li_r, li_d) and np.array_equal(hi_r, hi_d)
                                      and drawn_r == drawn_d)
    flt.synchronize()
    info = sharded.shard_info(flt)
    rows = np.r_[0:14, info.row_begin:info.row_end]
    S = flt.getFullSigma()
    res["mu"] = relf(flt.getFullState(), ref.mu)
    res["Sigma"] = relf(S[rows], ref.Sigma[rows])
    if ref64 is not None:
        res["mu64"] = relf(flt.getFullState(), ref64.mu)
        res["Sigma64"] = relf(S[rows], ref64.Sigma[rows])
        res["o32_mu64"] = relf(ref.mu, ref64.mu)                  # the fp32 ORACLE against the same truth
        res["o32_Sigma64"] = relf(ref.Sigma[rows], ref64.Sigma[rows])
    res["n_find"] = bool(np.array_equal(flt.featureIds()[1], [ft.n_find for ft in ref.features]))
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_hip_shard_update_flow_three_ranks_vs_oracle(dtype):
    """A sharded filter as a drop-in for the reference's update() (vR.cpp:964-1130 + 1245-1284): 2x2 St blocks and search
    ellipses, the full innovation covariance, the 1-point RANSAC hypotheses, the rescue gate, the device-list update and
    the two-stage composite (best hypothesis; the reference's seeded draw loop with the plane rows) on 3 ranks sharing
    the GPU -- every rank's answers against the oracle's."""
    world, n_feat = 3, 26
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gpu_flow_worker, args=(world, free_port(), n_feat, np.dtype(dtype).name, out), nprocs=world, join=True)
    f32 = dtype == np.float32
    for rank in range(world):
        r = out[rank]
        assert r["ellipses"] and r["counts"] and r["inliers"] == 0 and r["rescue"], (rank, r)
        assert r["two_stage_0"] and r["two_stage_1"] and r["n_find"], (rank, r)
        assert bound("rank S2x2 vs oracle", r["S2"], 2e-5 if f32 else 1e-10)
        assert bound("rank St vs oracle", r["St"], 2e-5 if f32 else 1e-10)
        # (fp32: measured 4.8e-5 -- the distance of the fp32 ORACLE from the truth, not of the HIP path: against the fp64
        # oracle below the same state is at 1.1e-5, and the yardstick line holds this one to 1.5 x |o32 - o64| + 2e-5)
        assert bound("rank mu vs oracle", r["mu"], 1e-4 if f32 else 1e-11)
        assert bound("rank Sigma rows vs oracle", r["Sigma"], 2e-3 if f32 else 1e-9)
        if f32:
            # Against the fp64 oracle run beside the fp32 one (same measurements, same inlier / rescue decisions): the
            # three-frame flow with 7 / 9 / 60 px outliers leaves the HIP fp32 path at the accuracy of every other
            # fp32 site; what the line above measures is the fp32 ORACLE's distance from the truth (explicit inverse,
            # (I - K H) Sigma), which the yardstick lines state
            assert r["decisions64"], (rank, r)
            print(f"flow fp32 rank {rank}: HIP-o64 mu {r['mu64']:.2e} Sigma {r['Sigma64']:.2e}; o32-o64 mu "
                  f"{r['o32_mu64']:.2e} Sigma {r['o32_Sigma64']:.2e}; HIP-o32 mu {r['mu']:.2e} Sigma {r['Sigma']:.2e}")
            assert bound("rank mu vs fp64 oracle", r["mu64"], 2e-5)
            assert bound("rank Sigma rows vs fp64 oracle", r["Sigma64"], 5e-4)
            assert r["mu"] <= 1.5 * r["o32_mu64"] + 2e-5, (rank, r)
            assert r["Sigma"] <= 1.5 * r["o32_Sigma64"] + 5e-4, (rank, r)


def _gpu_image_worker(rank, world, port, n_feat, out):
    """The image side on a SHARDED filter (Patch.cpp:215-293, libblur.cpp:17-79): frame, templates and the
    predicted blur replicated, the 2x2 gate blocks all-gathered, every rank searching every feature."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dataclasses
    import image_oracle as io_
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import sharded
    frame = io_.random_texture(240, 320, seed=24)
    moved = np.roll(np.roll(frame, 2, axis=1), -1, axis=0)
    gcfg = dict(pkg.kinect_config())
    gcfg.update(kernel_size=1000, T_camera=0.5)              # blur predictions are formed; the templates stay sharp (as in test_gpu_image.py)
    res = {}
    for tag in ("plain", "sharded"):
        f = pkg.VSlamFilter(gcfg, capacity_features=n_feat + 4, dtype=np.float32)
        f.setDt(1.0 / 30.0)
        f.setFrame(frame)
        for (u, v) in o.synthetic_pixels(o.Config.kinect(), n_feat):
            assert f.addFeature((u, v)) == 1
        full = f.getFullState()
        full[7:13] = [0.3, 0.0, 0.0, 0.0, 0.05, 0.0]       # the scenario of tests/test_gpu_image.py
        f.setFullState(full)
        if tag == "sharded":
            sharded.configure(f, rank, world)
        f.predict()
        hb = f.blurPredictions()
        f.setFrame(moved)
        z, found, score = f.findMatches()
        mp = np.stack([f.getPatch(i, matching=True) for i in range(0, n_feat, 3)])
        z2 = z[found.astype(bool)].reshape(-1)
        idx = np.nonzero(found)[0].astype(np.int32)
        f.update(z2.astype(np.float32), idx)                # the frame closes through the sharded update
        f.predict()
        f.setFrame(frame)
        zb, foundb, scoreb = f.findMatches()
        f.synchronize()
        res[tag] = (z, found, score, mp, hb, zb, foundb, scoreb)
        if tag == "sharded":
            res["info"] = sharded.shard_info(f).f_begin, sharded.shard_info(f).f_end
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_hip_shard_image_side_three_ranks_match_plain_filter():
    """ekf_set_frame / the predicted blur inside ekf_predict / ekf_find_matches on a sharded filter (3 ranks sharing the
    GPU): every rank's matches, scores, rewritten matching templates and blur predictions equal the plain filter's to the
    last bit in the first frame (same kernels on replicated inputs; the gated 2x2 blocks come from their owners), and
    after a sharded update + predict the second search still finds the same features."""
    world, n_feat = 3, 40
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gpu_image_worker, args=(world, free_port(), n_feat, out), nprocs=world, join=True)
    seen = set()
    for rank in range(world):
        r = out[rank]
        zp, fp, sp, mpp, hbp, zbp, fbp, sbp = r["plain"]
        zs, fs, ss, mps, hbs, zbs, fbs, sbs = r["sharded"]
        assert fp.sum() >= 20                                # the scenario does find most features
        assert np.array_equal(fs, fp) and np.array_equal(zs, zp) and np.array_equal(ss, sp), rank
        assert np.array_equal(mps, mpp) and np.array_equal(hbs, hbp), rank
        # second frame: the two filters went through different update code (sharded / plain): same matches, scores to rounding
        assert np.array_equal(fbs, fbp) and np.array_equal(zbs, zbp), rank
        assert np.allclose(sbs, sbp, rtol=0, atol=1e-4), rank
        seen.add(tuple(r["info"]))
    assert len(seen) == world                                # three different ownership ranges did run


@pytest.mark.gpu
@pytest.mark.parametrize("world,dist_chain", [(2, False), (3, False), (3, True)])
def test_rccl_smoke_script_rehearsal_over_gloo(world, dist_chain):
    """tools/rccl_smoke.py -- the one-command check a multi-GPU node runs over RCCL -- rehearsed here with `world` ranks
    sharing the GPU (gloo): N = 200 (ten 128-row tiles: the ranks' panels do NOT span the matrix), all-measured and
    subset + plane updates, a removal / addition, a conversion pass that shrinks n (zero padding checked on the device),
    the two-stage update; every rank against the plain path, fp64 to 1e-8 and fp32 to 5e-3.  Regression test of the
    round-3 finding: a converted feature's 3 x 3 cross blocks read the MIRROR feature's rows, which another rank owns.
    dist_chain (round 6): the same flow with the DISTRIBUTED chain forced on at this size (EKF_SHARD_DIST_MIN_BLOCKS=2: every
    update of two block steps or more -- all-measured, subset + plane, both stages of the two-stage update -- hands its panels
    round by all-gathers)."""
    import subprocess
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    if dist_chain:
        env["EKF_SHARD_DIST_MIN_BLOCKS"] = "2"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                        os.path.join(ROOT, "tools", "rccl_smoke.py"), "--backend", "gloo"],
                       capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith("[rccl_smoke]")]
    assert r.returncode == 0 and len(lines) == 2 and all(l.endswith("OK") for l in lines), (r.returncode, lines, r.stderr[-2000:])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_bench_ranks_command_rehearsal_over_gloo(world):
    """The exact command a SCALE driver runs for N = 2 (and, round 5, the same with 4 ranks: the box allows six processes on
    its GPU, and this test runner and the launcher are two of them; the 5-rank rehearsal is run outside pytest by
    tools/collect_profiles.sh, profiles/r5_bench_gloo_5ranks.json) -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 --steps 3 --warmup 1` -- with the ranks sharing the
    one GPU of the box and the collectives over gloo (EKF_BENCH_BACKEND; on a node the same code runs over RCCL):
    BASELINE configs[3] at full size (N = 1000), rank 0 prints ONE JSON line with the contract's keys, a finite state
    and the time of every all-gather (VERDICT r3 next #1e: the sharded downdate was rewritten after the last rehearsal)."""
    import json
    import subprocess
    env = dict(os.environ, EKF_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == world and doc["steps"] == 3 and doc["warmup"] == 1 and doc["run_sane"] is True
    assert doc["unit"] == "updates/s" and doc["value"] > 0 and doc["higher_is_better"] is True
    assert abs(doc["value"] * doc["ms_per_step"] * 1e-3 - 1.0) < 1e-2          # value = steps / the max-over-ranks time
    assert doc["config"]["features"] == 1000 and doc["config"]["backend"] == "gloo" and doc["scaling"] == "strong"
    ag = doc["allgather_ms_per_step"]
    assert {"H", "S", "V"} <= set(ag) and all(v > 0 for v in ag.values()), ag
    assert doc["jacobian_innovation_shard_ms"] > 0 and doc["roofline"]["frac"] > 0
    r0, r1 = doc["rank0_rows"]
    assert r0 == 14 and 14 < r1 < 14 + 6 * 1000                           # rank 0 owns the first part of the rows only


# BASELINE configs[4]: N = 4000, fp32, dynamic add / delete-feature covariance resize every 50 frames
N4000_FRAMES = 101            # resize after frames 49 and 99 (counting from 0), one more frame on the resized map


def n4000_cadence_run(pkg, flt, frames=N4000_FRAMES, every=50, n_feat=4000, progress=None):
    """The same calls for the plain and for the sharded filter: `frames` frames of the bench stream with every visible
    surviving feature measured; after every `every`-th frame 1 % of the features are removed (uniform indices, handed
    over in one call: the library removes in descending order as vR.cpp:1296 does) and as many are added at the end
    (vR.cpp:1300-1315), which are never measured (no stream behind them).  Returns mu, a sample of row indices and
    the invariants."""
    from ekf_monoslam_amd import synthetic
    cfg = pkg.kinect_config()
    px0, z = synthetic.measurement_stream(cfg, n_feat, frames, sigma_px=0.5)
    for (u, v) in px0:
        assert flt.addFeature((u, v)) == 1
    yield "built"
    rng = np.random.default_rng(1236)
    sid = np.arange(n_feat)                       # stream index behind every current feature, -1: none
    for k in range(frames):
        flt.predict()
        h, vis, rem, _ = flt.predictions()
        sel = np.nonzero(vis.astype(bool) & (sid >= 0))[0].astype(np.int32)
        flt.update(z[k][sid[sel]].reshape(-1), sel)
        if (k + 1) % every == 0:
            N = flt.numOfFeatures()
            drop = sorted(rng.choice(N, size=N // 100, replace=False).tolist())
            flt.removeFeatures(drop)
            sid = np.delete(sid, drop)
            for _ in range(len(drop)):
                assert flt.addFeature((float(rng.uniform(20, 300)), float(rng.uniform(20, 220)))) == 1
            sid = np.concatenate([sid, -np.ones(len(drop), np.int64)])
            assert flt.numOfFeatures() == N == len(sid)
        if progress and k % 10 == 9:
            progress(k)
    flt.synchronize()
    yield "done"


def _gpu_n4000_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import sharded
    N = 4000
    flt = pkg.VSlamFilter(pkg.kinect_config(), capacity_features=N + 64, dtype=np.float32)
    flt.setDt(1.0 / 30.0)
    run = n4000_cadence_run(pkg, flt, progress=(lambda k: print(f"[n4000 x{world}] frame {k + 1}", flush=True)) if rank == 0 else None)
    next(run)
    sharded.configure(flt, rank, world)
    next(run)
    info = sharded.shard_info(flt)
    pad, asym, big = flt.checkInvariants()
    rows = np.r_[0:14, info.row_begin:min(info.row_begin + 200, info.row_end), max(info.row_begin, info.row_end - 200):info.row_end]
    blocks = [flt.getSigmaBlock(int(r), 0, 1, flt.stateDim()) for r in rows[::7]]
    out[rank] = (flt.getFullState(), rows[::7], np.concatenate(blocks), (info.f_begin, info.f_end), pad, flt.numOfFeatures(),
                 info.rebalances, flt.launch_counts()["chain_dist_gather"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_hip_shard_n4000_resize_cadence_two_ranks_match_plain_path():
    """BASELINE configs[4] at FULL size and at its CADENCE (VERDICT r2 next #1d): N = 4000, n = 24 014, fp32, 63 block
    steps, 101 frames with 1 % of the features removed and as many added after every 50th frame, on two ranks sharing
    the GPU (collectives through gloo) -- every rank's rows against the plain single-GPU path on the same calls, the
    zero padding and the symmetry of the plain path's covariance checked on the device after the last frame.
    (The fp64 oracle cannot follow at this size: one dense update is 9 TFLOP and Sigma 4.6 GB.)"""
    from __graft_entry__ import load_package
    pkg = load_package()
    N = 4000
    flt = pkg.VSlamFilter(pkg.kinect_config(), capacity_features=N + 64, dtype=np.float32)
    flt.setDt(1.0 / 30.0)
    # the same column chunks as the two-rank step (4 equal chunks; the plain default is 3 / 8 / 16 sixteenths): every
    # tile product then sums the same terms in the same order on both paths, and the comparison below is EXACT.  (With
    # different chunk plans the two fp32 runs drift apart by 1e-5 .. 1e-3 over 100 frames -- rounding differences of
    # 1e-7 in V amplified by the filter, tools/shard_cadence_probe.py -- which would say nothing about the sharding.)
    flt.set_option(3, 4)
    # (round 5: both paths run their defaults otherwise -- the covariance downdate on the bf16 matrix pipe, where every
    # element pair of Sigma is ONE sum whichever rank computes it (k_syrk_bf16x6), and the sequential form of the chunked
    # update, W of the next chunk re-evaluated from the downdated rows, on the ranks as on the plain path)
    for _ in n4000_cadence_run(pkg, flt, progress=lambda k: print(f"[n4000 plain] frame {k + 1}", flush=True)):
        pass
    mu_p = flt.getFullState()
    n = flt.stateDim()
    pad, asym, big = flt.checkInvariants()
    assert pad == 0.0 and asym == 0.0 and np.isfinite(big)
    assert np.all(np.isfinite(mu_p)) and abs(np.linalg.norm(mu_p[3:7]) - 1.0) < 1e-5
    d = np.diag(flt.getSigmaBlock(0, 0, 2000, 2000))
    assert np.all(d > 0)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gpu_n4000_worker, args=(2, free_port(), out), nprocs=2, join=True)
    ranges = []
    for rank in range(2):
        mu, rows, S_rows, frange, pad, nfeat, rebal, dist_gathers = out[rank]
        assert nfeat == N and mu.shape == (n,) and np.all(np.isfinite(mu)) and pad == 0.0
        # round 6: at this size (~60 block steps) the ranks run the DISTRIBUTED chain (one all-gather of the panel per block
        # step, Filter::dist_chain_steps); the plain path it is compared with factors S in one place
        assert dist_gathers > 50 * N4000_FRAMES, dist_gathers
        ref_rows = np.concatenate([flt.getSigmaBlock(int(r), 0, 1, n) for r in rows])
        # bit-identical after 101 frames and two resizes -- or the ONE documented deviation of this rig (ranks sharing a
        # GPU: an anchor coordinate off by < 1e-7 about once in 400 updates, helpers.exact_or_anchor_glitch)
        exact_or_anchor_glitch(f"rank {rank} (rows {rows[14]} .. {rows[-1]})", mu, mu_p, S_rows, ref_rows, rows)
        ranges.append(frange)
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == N


@pytest.mark.gpu
def test_shard_argument_checks():
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import sharded
    flt = pkg.VSlamFilter(pkg.kinect_config(), capacity_features=8)
    for (u, v) in o.synthetic_pixels(o.Config.kinect(), 5):
        flt.addFeature((u, v))
    lib = flt._lib
    assert lib.ekf_shard_configure(flt._h, 2, 2, None, None) == 1          # rank out of range
    assert lib.ekf_shard_configure(flt._h, 0, 2, None, None) == 1          # world > 1 without a callback
    assert lib.ekf_shard_rebalance(flt._h) == 4                            # not configured
    sharded.configure(flt, 0, 1)
    info = sharded.shard_info(flt)
    assert (info.f_begin, info.f_end, info.row_begin, info.row_end) == (0, 5, 14, 44)
    with pytest.raises(pkg.EkfError):                                      # update before predict
        flt.update(np.zeros(2, np.float32), [0])
    flt.predict()
    with pytest.raises(pkg.EkfError):                                      # unsorted list
        flt.update(np.zeros(4, np.float32), [3, 1])
