"""Row-panel sharding of the EKF step across the GPUs of one node, one process per GPU
(include/ekf_monoslam.h "multi-GPU").

Every rank holds the same filter (same calls in the same order everywhere) and OWNS a contiguous range of
features: the rows of Sigma of those features (all columns) plus a replica of the camera rows stay valid on it; mu
is replicated.  A step exchanges only DISJOINT panels, each by ONE all-gather of equal-sized staging slots (no
reduction anywhere: on a fully connected xGMI node every peer pair moves its panel over its own link):

    predict                  -> per-feature records h | Hc | Hf | flags           ("reassemble H")
    update: W rows, S rows   -> row panels of S                                   ("reassemble S")
            per column chunk of the factorisation, beside the Cholesky chain (replicated; from 40 block steps on
            distributed: cyclic row blocks, one all-gather of the own panel blocks per block step):
            V_g = W_g Z_gg   -> own rows of V_g                                    (n x 2M scalars per step in all)
            Sigma[own rows] -= V_g[own rows] V_g^T
    convert2XYZ_ifLinearAll  -> linearity flags (one byte per feature)
    re-balance               -> row panels of Sigma (only when a rank owns > 1.125 x the mean rows)

The orchestration lives in the library (csrc/ekf_capi.hip: shard_predict / shard_update / ...): it packs, calls the
host's all-gather on its own stream, and unpacks.  This module supplies that callback over `torch.distributed`
(backend "nccl" = RCCL on GPUs; "gloo" with host staging when several ranks share one GPU in the tests) and the
multi-GPU leg of bench.py.  (The numpy model of the same protocol that the CPU tests run over gloo lives with the tests:
tests/shard_protocol.py.)
"""
from __future__ import annotations

import ctypes as C
import os
import time
import traceback

import numpy as np

ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class ShardInfo(C.Structure):
    """struct ekf_shard_info."""
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("N", C.c_int), ("state_dim", C.c_int),
                ("f_begin", C.c_int), ("f_end", C.c_int), ("row_begin", C.c_int), ("row_end", C.c_int),
                ("max_rows_any_rank", C.c_int), ("rebalances", C.c_int)]


class _DevArray:
    """Minimal __cuda_array_interface__ carrier: lets torch view library memory without a copy."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2}


class TorchAllGather:
    """The ekf_allgather_fn of a rank: `world` equal slots, slot g = rank g's send buffer, on the library's stream.

    nccl (RCCL): `all_gather_into_tensor(recv, send)` on plain, contiguous, separate send / receive staging buffers
    -- the most ordinary form of the collective; torch enqueues it behind the current stream (the library's stream,
    made current through `ExternalStream`) and makes that stream wait for it.
    gloo (several ranks on one GPU, tests): the slots travel through host memory."""

    def __init__(self, world, rank, device_index=0):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world, self.rank, self.device_index = world, rank, device_index
        self.backend = dist.get_backend() if dist.is_initialized() else "none"
        self._views = {}
        self._streams = {}
        self._groups = {}                       # nccl: one process group (= one communicator) per library stream
        self.calls = 0
        self.bytes_sent = 0
        self.seconds = 0.0                      # host time inside the callback (gloo: the whole exchange)
        self.error = None
        self.c_callback = ALLGATHER_FN(self._call)

    def _view(self, ptr, nbytes):
        key = (ptr, nbytes)
        v = self._views.get(key)
        if v is None:
            if len(self._views) > 64:
                self._views.clear()
            v = self.torch.as_tensor(_DevArray(ptr, nbytes), device=f"cuda:{self.device_index}")
            self._views[key] = v
        return v

    def _call(self, ctx, d_send, d_recv, nbytes, stream):
        try:
            t0 = time.perf_counter()
            torch, dist = self.torch, self.dist
            send = self._view(d_send, nbytes)
            recv = self._view(d_recv, nbytes * self.world)
            st = self._streams.get(stream)
            if st is None:
                st = torch.cuda.ExternalStream(stream, device=f"cuda:{self.device_index}") if stream else \
                    torch.cuda.default_stream(self.device_index)
                self._streams[stream] = st
            if stream not in self._groups:
                # The library gathers on two streams at once (main stream: H, S and -- round 6 -- the panels of the
                # distributed chain; gather stream: the rows of V_g).  Collectives of ONE communicator run in issue order,
                # which would make the chain's next panel wait for the solve + gather of the chunk before; a communicator per
                # stream keeps the two sequences independent.  Every rank meets a new stream at the same call, so the
                # (collective) creation of the group is in step.
                self._groups[stream] = None if (not self._groups or self.backend != "nccl") else dist.new_group(backend="nccl")
            with torch.cuda.stream(st):
                if self.backend == "nccl":
                    dist.all_gather_into_tensor(recv, send, group=self._groups[stream])
                else:
                    host = send.cpu()
                    parts = [torch.empty_like(host) for _ in range(self.world)]
                    dist.all_gather(parts, host)
                    recv.copy_(torch.cat(parts))
            self.calls += 1
            self.bytes_sent += int(nbytes)
            self.seconds += time.perf_counter() - t0
            return 0
        except Exception as e:                  # never let an exception unwind through the C frames
            self.error = e
            traceback.print_exc()
            return 1


def configure(flt, rank, world, device_index=0):
    """Switch a VSlamFilter (every rank built it by the same calls) to sharded operation over torch.distributed.
    EKF_SHARD_FORCE_COLLECTIVE=1 keeps the exchanges at world 1 too (the collective path on one GPU, for profiling)."""
    force = os.environ.get("EKF_SHARD_FORCE_COLLECTIVE", "0") not in ("", "0")
    ag = TorchAllGather(world, rank, device_index) if (world > 1 or force) else None
    flt._allgather = ag                          # keeps the ctypes callback alive as long as the filter
    flt._check(flt._lib.ekf_shard_configure(flt._h, int(rank), int(world),
                                            ag.c_callback if ag else C.cast(None, ALLGATHER_FN), None))
    return ag


def shard_info(flt):
    info = ShardInfo()
    flt._check(flt._lib.ekf_shard_get_info(flt._h, C.byref(info)))
    return info


def shard_update(flt, d_z_ptr, indices, plane=False):
    idx = np.ascontiguousarray(indices, np.int32)
    flt._check(flt._lib.ekf_shard_update(flt._h, C.c_void_p(d_z_ptr), idx.ctypes.data_as(C.c_void_p), idx.size,
                                         int(bool(plane))))


def rebalance(flt):
    flt._check(flt._lib.ekf_shard_rebalance(flt._h))


# ---------------------------------------------------------------------------------------------
# bench.py --gpus N (N > 1): one rank per GPU over RCCL
# ---------------------------------------------------------------------------------------------
def bench(pkg, cfg, n_feat, px0, z, args, rank, world, dev):
    import torch
    import torch.distributed as dist
    import bench as _bench
    # a map runs `seg` frames (bench.segment_frames: the run lengths over which the fp32 covariance was VERIFIED to stay
    # positive with every feature measured in every frame); longer runs continue on a map started
    # afresh from the stream's current pixels -- all maps are built before the clock starts
    seg = _bench.segment_frames(n_feat)
    frames = args.warmup + args.steps
    nseg = max(1, -(-frames // seg))
    if nseg > 64:
        raise SystemExit(f"--steps {frames}: more than 64 map restarts of {seg} frames at N = {n_feat}; use fewer steps")
    maps = []
    for sgi in range(nseg):
        f_ = pkg.VSlamFilter(cfg, capacity_features=n_feat, dtype=np.float32, device=dev.index)
        f_.setDt(1.0 / 30.0)
        for (u, v) in (px0 if sgi == 0 else z[sgi * seg - 1]):
            if f_.addFeature((u, v)) != 1:
                raise RuntimeError("synthetic pixel rejected by addFeature")
        f_.synchronize()
        configure(f_, rank, world, dev.index)
        maps.append(f_)
    flt = maps[0]
    d_z = torch.from_numpy(z.reshape(z.shape[0], -1)).to(dev).contiguous()
    idx = np.arange(n_feat, dtype=np.int32)
    bpf = 2 * n_feat * 4
    n = flt.stateDim()

    def run(first, count):
        for f in range(first, first + count):
            m_ = maps[min(f // seg, nseg - 1)]
            m_.predict()
            shard_update(m_, d_z.data_ptr() + f * bpf, idx)

    def sync_all():
        for m_ in maps:
            m_.synchronize()
        torch.cuda.synchronize()

    run(0, args.warmup)
    sync_all()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    sync_all()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                           device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    mu = maps[min((frames - 1) // seg, nseg - 1)].getFullState()
    sane = bool(np.all(np.isfinite(mu)) and abs(np.linalg.norm(mu[3:7]) - 1) < 1e-4)
    # per-phase share of one step on this rank (HIP events around every kernel and every exchange, separate short pass)
    flt.set_option(2, 2)
    flt.profile_reset()
    k = min(5, args.steps)
    run(args.warmup, k)
    sync_all()
    prof = flt.profile()
    work = flt.profile_work()
    flt.set_option(2, 0)
    phase = {name: round(ms / k, 4) for name, (ms, cnt) in prof.items()}
    gather_ms = {name[len("allgather_"):].upper(): v for name, v in phase.items() if name.startswith("allgather_")}
    shard_ms = sum(phase.get(x, 0.0) for x in ("measure", "innovation", "sigma_ht", "innovation_cov"))
    # dominant kernel of a rank: its row panel of the downdate, rows x n x m multiply-adds, every column
    # (a row panel cannot use the symmetry), timed with HIP events on the library's streams in the pass above
    dd_ms, dd_cnt = prof.get("downdate_syrk", (0.0, 0))
    roofline = None
    if dd_cnt:
        flop = work.get("downdate_syrk", 0.0) / dd_cnt
        ach = flop / (dd_ms / dd_cnt * 1e-3) / 1e12
        split = flt.launch_counts().get("downdate_bf16x6", 0) > 0     # which downdate kernel the rank RAN (ekf_launch_count), not a re-derived rule
        if split:
            # k_syrk_bf16x6: six bf16 products per algorithmic fp32 product -> roofline = dense bf16 peak / 6 (bench.py)
            peak6 = 2500.0 / 6.0
            roofline = {"kernel": "downdate of the rank's canonical tiles (k_syrk_bf16x6: v_mfma_f32_32x32x16_bf16, 3 x bf16 split "
                                  "operands, six products per fp32 product), rank 0", "bound": "mfma",
                        "achieved": round(ach, 2), "peak": round(peak6, 1), "unit": "TFLOP/s (algorithmic fp32 flop)",
                        "frac": round(ach / peak6, 4), "vs_f32_mfma_peak": round(ach / 157.3, 4),
                        "traffic": None, "avg_launch_ms": round(dd_ms / dd_cnt, 4),
                        "algorithmic_flop_per_launch": flop, "launches_per_step": dd_cnt / k}
        else:
            roofline = {"kernel": "downdate row panel (k_gemm_mfma, f32 MFMA 32x32x2), rank 0", "bound": "mfma",
                        "achieved": round(ach, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(ach / 157.3, 4),
                        "traffic": None, "avg_launch_ms": round(dd_ms / dd_cnt, 4),
                        "algorithmic_flop_per_launch": flop, "launches_per_step": dd_cnt / k}
    info = shard_info(flt)
    dist_gathers = flt.launch_counts().get("chain_dist_gather", 0)       # (which form of the chain the library took)
    result = {
        "metric": "EKF updates/sec at N features (state dim 14+6N)",
        "value": round(args.steps / elapsed, 2), "unit": "updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32" if roofline is None or "bf16x6" not in roofline["kernel"] else
                 "f32 (covariance downdate products as 3 x bf16 splits of the fp32 operands, six products per fp32 product, "
                 "fp32 accumulate; EKF_SPLIT_BF16=0: v_mfma_f32_32x32x2_f32)",
        "data": "synthetic",
        "config": {"workload": f"N={n_feat} inverse-depth features, n={n}, M=N measured per frame, fp32, "
                               f"row-panel shard over {world} GPUs (BASELINE configs[3])",
                   "features": n_feat, "state_dim": n, "measured_per_frame": n_feat,
                   "parallelism": f"row-panel shard x{world}: all-gather H, S, V over "
                                  f"{'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend() + ' (host staging, functional rehearsal)'}"
                                  + (", chunk-pipelined beside the DISTRIBUTED Cholesky chain (cyclic row blocks, one all-gather "
                                     "of the panel per block step)" if dist_gathers > 0 else
                                     ", chunk-pipelined beside the replicated Cholesky chain"),
                   "backend": dist.get_backend(),
                   "frames_per_map": seg, "maps": nseg},
        "run_sane": sane,
        "rank0_rows": [info.row_begin, info.row_end],
        "per_rank_kernel_ms": phase,
        "jacobian_innovation_shard_ms": round(shard_ms, 4),
        "allgather_ms_per_step": gather_ms,
        "roofline": roofline, "cpu_baseline": None,
    }
    for f_ in maps:
        f_.close()
    return result
