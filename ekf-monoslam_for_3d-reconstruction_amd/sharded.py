"""Row-panel sharding of the EKF step across the GPUs of one node, one process per GPU
(include/ekf_monoslam.h "multi-GPU").

Every rank holds the same feature list.  Rank g owns features [N g/G, N (g+1)/G), keeps the
rows of Sigma of those features (all columns) plus a replica of the camera rows up to date, and
a step is four local phases separated by all-gathers of DISJOINT panels (no reduction anywhere:
on a fully connected xGMI node every peer pair moves its panel over its own link):

    predict        -> all-gather h, Hc, Hf, flags     (per-feature slices, ~128 B / feature)
    innovation     -> all-gather row panels of S      ((2M)^2 s bytes in total)
    factor_solve   -> all-gather row panels of V      (n 2M s bytes in total)
    downdate

`ShardedStep` only orchestrates; the arithmetic is in the backend it is given: `HipShardBackend`
(the C ABI on a GPU, tensors are zero-copy views of the library's buffers) or, in the CPU tests,
an oracle-backed stand-in with the same phase methods.  Collectives go through
`torch.distributed` (backend "nccl" = RCCL on GPUs; "gloo" in the CPU tests).
"""
from __future__ import annotations

import ctypes as C
import json
import os
import time

import numpy as np


class ShardView(C.Structure):
    """struct ekf_shard_view."""
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("N", C.c_int), ("f_begin", C.c_int), ("f_end", C.c_int),
                ("camera_dim", C.c_int), ("rows_per_rank", C.c_int),
                ("m", C.c_int), ("m_pad", C.c_int), ("ldy", C.c_int),
                ("d_h", C.c_void_p), ("d_Hc", C.c_void_p), ("d_Hf", C.c_void_p), ("d_flags", C.c_void_p),
                ("d_S", C.c_void_p), ("d_V", C.c_void_p)]


class _DevArray:
    """Minimal __cuda_array_interface__ carrier: lets torch view library memory without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def all_gather_rows(t, start, count, rank, world):
    """In-place all-gather of `world` disjoint row blocks of the 2-D tensor `t`: rank r contributes
    rows [start + r*count, start + (r+1)*count) and receives everybody else's."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return
    region = t[start:start + world * count]
    own = t[start + rank * count:start + (rank + 1) * count]
    backend = dist.get_backend()
    if backend == "nccl" and region.is_cuda:
        dist.all_gather_into_tensor(region.reshape(-1), own.reshape(-1))     # in place: own is a slice of region
        return
    # gloo (CPU tests, or several ranks sharing one GPU): stage through host copies
    src = own.detach().cpu().contiguous()
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src)
    for r, p in enumerate(parts):
        if r != rank:
            t[start + r * count:start + (r + 1) * count].copy_(p)


class HipShardBackend:
    """Phase methods of one rank on its GPU (the ekf_shard_* entry points)."""

    def __init__(self, flt, rank, world, stream=None):
        import torch
        self.flt = flt
        self.lib = flt._lib
        self.rank, self.world = rank, world
        self._check(self.lib.ekf_shard_configure(flt._h, rank, world))
        # the phases and the collectives must be ordered on ONE stream: run the library on torch's
        # current stream (the stream RCCL synchronises with) unless the caller names another one
        flt.set_stream(torch.cuda.current_stream().cuda_stream if stream is None else stream)
        self.torch = torch
        self.dtype = torch.float32 if flt.dtype == np.float32 else torch.float64
        self.typestr = "<f4" if flt.dtype == np.float32 else "<f8"
        self._views = None
        self.refresh_view()

    def _check(self, rc):
        self.flt._check(rc)

    def refresh_view(self):
        v = ShardView()
        self._check(self.lib.ekf_shard_get_view(self.flt._h, C.byref(v)))
        self.view = v
        self.N, self.f0, self.f1 = v.N, v.f_begin, v.f_end
        self.camera_dim, self.rows_per_rank, self.ldy = v.camera_dim, v.rows_per_rank, v.ldy
        return v

    def tensors(self):
        """Zero-copy torch views: h (N,2), Hc (N,14), Hf (N,12), flags (N,1), S (2N,ldy), V (n,ldy)."""
        if self._views is None:
            t, v = self.torch, self.view
            n = self.camera_dim + 6 * self.N

            def view(ptr, shape, ts):
                return t.as_tensor(_DevArray(ptr, shape, ts), device="cuda")
            self._views = {
                "h": view(v.d_h, (self.N, 2), self.typestr), "Hc": view(v.d_Hc, (self.N, 14), self.typestr),
                "Hf": view(v.d_Hf, (self.N, 12), self.typestr), "flags": view(v.d_flags, (self.N, 1), "|u1"),
                "S": view(v.d_S, (2 * self.N, self.ldy), self.typestr), "V": view(v.d_V, (n, self.ldy), self.typestr)}
        return self._views

    def predict(self):
        self._check(self.lib.ekf_shard_predict(self.flt._h, None, None, 0))

    def innovation(self, d_z_ptr, M):
        self._check(self.lib.ekf_shard_innovation(self.flt._h, C.c_void_p(d_z_ptr), int(M), 0))

    def factor_solve(self):
        self._check(self.lib.ekf_shard_factor_solve(self.flt._h))

    def downdate(self):
        self._check(self.lib.ekf_shard_downdate(self.flt._h))


class ShardedStep:
    """One EKF step (predict + full-batch update over M = N features) across the ranks."""

    def __init__(self, backend):
        self.b = backend
        self.rank, self.world = backend.rank, backend.world
        self.comm_s = 0.0

    def _gather(self, what, t, start, count):
        """One all-gather; with `self.timing` set (a dict of event-pair lists) it is bracketed by events on
        the current stream, the stream the phases and RCCL are ordered on."""
        if self.timing is None:
            all_gather_rows(t, start, count, self.rank, self.world)
            return
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        all_gather_rows(t, start, count, self.rank, self.world)
        e1.record()
        self.timing.setdefault(what, []).append((e0, e1))

    timing = None

    def step(self, z):
        """z: backend-specific handle of this frame's 2N measurements (device pointer for the HIP
        backend, array for the CPU stand-in)."""
        b = self.b
        nf = b.f1 - b.f0
        b.predict()
        ts = b.tensors()
        for name in ("h", "Hc", "Hf", "flags"):                     # reassemble H
            self._gather("H", ts[name], 0, nf)
        b.innovation(z, b.N)
        self._gather("S", ts["S"], 0, 2 * nf)                        # reassemble S
        b.factor_solve()
        self._gather("V", ts["V"], b.camera_dim, b.rows_per_rank)
        b.downdate()


# ---------------------------------------------------------------------------------------------
# bench.py --gpus N (N > 1): one rank per GPU over RCCL
# ---------------------------------------------------------------------------------------------
def bench(pkg, cfg, n_feat, px0, z, args, rank, world, dev):
    import torch
    import torch.distributed as dist
    if n_feat % world != 0:
        raise SystemExit(f"--features {n_feat} must be divisible by the number of GPUs ({world})")
    # a map runs `seg` frames (bench.segment_frames: the fp32 covariance of a map whose features are ALL measured in
    # EVERY frame stops being positive after a few hundred frames); longer runs continue on a map started afresh
    # from the stream's current pixels -- all maps are built before the clock starts
    import bench as _bench
    seg = _bench.segment_frames(n_feat)
    frames = args.warmup + args.steps
    nseg = max(1, -(-frames // seg))
    if nseg > 64:
        raise SystemExit(f"--steps {frames}: more than 64 map restarts of {seg} frames at N = {n_feat}; use fewer steps")
    # everything of the step -- library phases and collectives -- is ordered on ONE non-default stream: on the legacy
    # default stream every kernel would synchronise with the library's internal second stream
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    maps = []
    for sgi in range(nseg):
        f_ = pkg.VSlamFilter(cfg, capacity_features=n_feat, dtype=np.float32, device=dev.index)
        f_.setDt(1.0 / 30.0)
        for (u, v) in (px0 if sgi == 0 else z[sgi * seg - 1]):
            if f_.addFeature((u, v)) != 1:
                raise RuntimeError("synthetic pixel rejected by addFeature")
        f_.synchronize()
        b_ = HipShardBackend(f_, rank, world, stream=torch.cuda.current_stream().cuda_stream)
        maps.append((f_, ShardedStep(b_)))
    flt, stepper = maps[0]
    d_z = torch.from_numpy(z.reshape(z.shape[0], -1)).to(dev).contiguous()
    bpf = 2 * n_feat * 4
    n = flt.stateDim()

    def run(first, count):
        for f in range(first, first + count):
            maps[min(f // seg, nseg - 1)][1].step(d_z.data_ptr() + f * bpf)

    run(0, args.warmup)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                           device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    mu = maps[min((frames - 1) // seg, nseg - 1)][0].getFullState()
    sane = bool(np.all(np.isfinite(mu)) and abs(np.linalg.norm(mu[3:7]) - 1) < 1e-4)
    # per-phase share of one step on this rank (HIP events around every kernel, separate short pass)
    flt.set_option(2, 2)
    flt.profile_reset()
    stepper.timing = {}
    run(args.warmup, min(5, args.steps))
    torch.cuda.synchronize()
    prof = flt.profile()
    flt.set_option(2, 0)
    k = min(5, args.steps)
    gather_ms = {what: round(sum(a.elapsed_time(b) for a, b in pairs) / k, 4) for what, pairs in stepper.timing.items()}
    stepper.timing = None
    phase = {name: round(ms / k, 4) for name, (ms, cnt) in prof.items()}
    shard_ms = sum(phase.get(x, 0.0) for x in ("measure", "sigma_ht", "innovation_cov"))
    # dominant kernel of a rank: its row panel of the downdate, (n / G) x n x m multiply-adds, every column
    # (a panel cannot use the symmetry), timed with HIP events on the library's stream in the pass above
    dd_ms, dd_cnt = prof.get("downdate_syrk", (0.0, 0))
    roofline = None
    if dd_cnt:
        launches_per_step = dd_cnt / k
        rows = n / world
        flop = 2.0 * rows * n * (2 * n_feat) / max(1.0, round(launches_per_step))
        ach = flop / (dd_ms / dd_cnt * 1e-3) / 1e12
        roofline = {"kernel": "downdate row panel (k_gemm_nt_mfma, f32 MFMA 32x32x2), rank 0", "bound": "mfma",
                    "achieved": round(ach, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(ach / 157.3, 4),
                    "traffic": None, "avg_launch_ms": round(dd_ms / dd_cnt, 4),
                    "algorithmic_flop_per_launch": flop, "launches_per_step": launches_per_step}
    result = {
        "metric": "EKF updates/sec at N features (state dim 14+6N)",
        "value": round(args.steps / elapsed, 2), "unit": "updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"N={n_feat} inverse-depth features, n={n}, M=N measured per frame, fp32, "
                               f"row-panel shard over {world} GPUs (BASELINE configs[3])",
                   "features": n_feat, "state_dim": n, "measured_per_frame": n_feat,
                   "parallelism": f"row-panel shard x{world}: all-gather H, S, V over RCCL",
                   "frames_per_map": seg, "maps": nseg},
        "run_sane": sane,
        "per_rank_kernel_ms": phase,
        "jacobian_innovation_shard_ms": round(shard_ms, 4),
        "allgather_ms_per_step": gather_ms,
        "roofline": roofline, "cpu_baseline": None,
    }
    for f_, _ in maps:
        f_.close()
    return result
