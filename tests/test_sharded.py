"""The N > 1 path: row-panel sharding with all-gathers of H, S and V.

CPU (gloo, world_size 2 and 4): the real orchestration (`ShardedStep`, `all_gather_rows`) over an
oracle-backed stand-in that poisons every row a rank does not own -> must equal the unsharded
oracle.  GPU: world 1 in-process against the plain HIP path, and two ranks sharing the one GPU of
the box over gloo against the plain HIP path."""
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


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


from helpers import bound, relf  # noqa: E402


def reference_run(n_feat, frames, dtype):
    f = o.build_scenario(o.StructuredFilter, o.Config.kinect(), n_feat, dtype)
    zs = []
    for k in range(frames):
        f.predict()
        idx = list(range(n_feat))
        z = o.synthetic_measurements(f, idx, seed=500 + k, sigma=0.5)
        zs.append(z)
        f.update(z, idx)
    return f, zs


def _cpu_worker(rank, world, port, n_feat, frames, zs, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    load_package()
    from ekf_monoslam_amd.sharded import ShardedStep
    from sharded_common import OracleShardBackend
    f = o.build_scenario(o.StructuredFilter, o.Config.kinect(), n_feat, np.float64)
    step = ShardedStep(OracleShardBackend(f, rank, world))
    for k in range(frames):
        step.step(zs[k])
    rows = step.b.own_rows()
    out[rank] = (f.mu.copy(), rows, f.Sigma[rows].copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_orchestration_matches_unsharded_oracle_gloo(world):
    n_feat, frames = 8, 3
    ref, zs = reference_run(n_feat, frames, np.float64)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_cpu_worker, args=(world, free_port(), n_feat, frames, zs, out), nprocs=world, join=True)
    seen = np.zeros(ref.n, bool)
    for rank in range(world):
        mu, rows, S_rows = out[rank]
        assert bound("mu, ref.mu", relf(mu, ref.mu), 1e-9)
        assert bound("S_rows, ref.Sigma[rows]", relf(S_rows, ref.Sigma[rows]), 1e-8)
        seen[rows] = True
    assert seen.all()                      # the panels cover every row of Sigma


def _mk_hip(pkg, n_feat, px0=None):
    """HIP filter on the scenario of o.build_scenario (velocities set, features added in order), or -- with
    px0 -- on the bench's synthetic stream (camera at rest, its pixels)."""
    cfg = o.Config.kinect()
    flt = pkg.VSlamFilter(pkg.kinect_config(), capacity_features=n_feat, dtype=np.float32)
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


def _gpu_worker(rank, world, port, n_feat, frames, z_np, out, px0=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd.sharded import HipShardBackend, ShardedStep
    flt = _mk_hip(pkg, n_feat, px0)
    b = HipShardBackend(flt, rank, world)
    step = ShardedStep(b)
    d_z = torch.from_numpy(z_np).cuda()
    for k in range(frames):
        step.step(d_z[k].data_ptr())
        flt.synchronize()
    mu = flt.getFullState()
    r0 = b.camera_dim + rank * b.rows_per_rank
    rows = np.r_[0:b.camera_dim, r0:r0 + b.rows_per_rank]
    S = flt.getFullSigma()
    out[rank] = (mu, rows, S[rows])
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


def _stream(n_feat, frames):
    ref = o.build_scenario(o.StructuredFilter, o.Config.kinect(), n_feat, np.float32)
    zs = []
    for k in range(frames):
        ref.predict()
        z = o.synthetic_measurements(ref, list(range(n_feat)), seed=700 + k, sigma=0.5)
        zs.append(z)
        ref.update(z, list(range(n_feat)))
    return ref, np.stack(zs).astype(np.float32)


@pytest.mark.gpu
def test_hip_shard_world1_matches_plain_path():
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd.sharded import HipShardBackend, ShardedStep
    n_feat, frames = 40, 3
    ref, z_np = _stream(n_feat, frames)
    mu_p, S_p = _plain_hip_run(n_feat, frames, z_np)
    flt = _mk_hip(pkg, n_feat)
    step = ShardedStep(HipShardBackend(flt, 0, 1))
    d_z = torch.from_numpy(z_np).cuda()
    for k in range(frames):
        step.step(d_z[k].data_ptr())
    flt.synchronize()
    assert bound("flt.getFullState(), mu_p", relf(flt.getFullState(), mu_p), 1e-5)
    assert bound("flt.getFullSigma(), S_p", relf(flt.getFullSigma(), S_p), 2e-4)
    assert bound("flt.getFullSigma(), ref.Sigma", relf(flt.getFullSigma(), ref.Sigma), 1e-3)


@pytest.mark.gpu
def test_hip_shard_two_ranks_on_one_gpu_match_plain_path():
    n_feat, frames = 40, 3
    ref, z_np = _stream(n_feat, frames)
    mu_p, S_p = _plain_hip_run(n_feat, frames, z_np)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gpu_worker, args=(2, free_port(), n_feat, frames, z_np, out), nprocs=2, join=True)
    for rank in range(2):
        mu, rows, S_rows = out[rank]
        assert bound("mu, mu_p", relf(mu, mu_p), 1e-5)
        assert bound("S_rows, S_p[rows]", relf(S_rows, S_p[rows]), 2e-4)


@pytest.mark.gpu
def test_hip_shard_four_ranks_mid_size_match_plain_path():
    """Row panels that do not sit on tile boundaries (150 features = 900 rows per rank), a chain of 10 block
    steps in chunks, four ranks sharing the GPU over gloo: every rank's rows must match the plain pipelined path."""
    n_feat, frames, world = 600, 2, 4
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import synthetic
    px0, z = synthetic.measurement_stream(pkg.kinect_config(), n_feat, frames, sigma_px=0.5)
    z_np = np.ascontiguousarray(z.reshape(frames, -1), np.float32)
    mu_p, S_p = _plain_hip_run(n_feat, frames, z_np, px0)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gpu_worker, args=(world, free_port(), n_feat, frames, z_np, out, px0), nprocs=world, join=True)
    for rank in range(world):
        mu, rows, S_rows = out[rank]
        assert np.all(np.isfinite(mu))
        assert bound("mu, mu_p", relf(mu, mu_p), 2e-5)
        assert bound("S_rows, S_p[rows]", relf(S_rows, S_p[rows]), 5e-4)


@pytest.mark.gpu
def test_shard_rejects_unsupported_layouts():
    from __graft_entry__ import load_package
    pkg = load_package()
    flt = pkg.VSlamFilter(pkg.kinect_config(), capacity_features=8)
    for (u, v) in o.synthetic_pixels(o.Config.kinect(), 5):
        flt.addFeature((u, v))
    assert flt._lib.ekf_shard_configure(flt._h, 0, 2) == 6        # 5 features over 2 ranks
    assert flt._lib.ekf_shard_configure(flt._h, 0, 5) == 0
    assert flt._lib.ekf_shard_factor_solve(flt._h) == 4            # phase order is enforced
