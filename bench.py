#!/usr/bin/env python3
"""EKF-updates/sec of the MI355X-native predict/update core on a synthetic N-feature map.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on
rank 0.  A "step" is one EKF update = predict (covariance propagation + state) + h/H for all
N features + one full-batch update over M = N measured features, with the
measurement stream and the index list already resident in HBM.  Workload at N=1 is
BASELINE.json configs[2] (N = 1000 inverse-depth features, n = 6014, fp32), the configuration
the north-star target is quoted on; `--features 200` runs configs[1].

Extra objects on the same line:
  roofline      dominant kernel (the symmetric downdate Sigma -= V V^T on f32 MFMA), algorithmic
                flop / HIP-event duration measured over the timed steps.
  p_propagate   HBM GB/s of the streaming P <- F P F^T + Q kernel (2 n^2 s bytes), measured in a
                second pass of the same steps with EKF_OPT_PROPAGATE_STREAMING.
  cpu_baseline  the reference's dense formulation restated in numpy/OpenBLAS (oracle, "port"),
                one frame of the same workload on the host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_F32_MFMA_TF = 157.3     # MI355X_MICROARCH.md: f32-input MFMA = vector peak
PEAK_BF16_MFMA_TF = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (~2.5 PF; the 5 PF headline includes 2:1 sparsity)
SIGMA_Z_PX = 0.5              # pixel noise of the synthetic stream (the filter's R stays sigma_pixel^2 = 4)


def csrc_sha():
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "ekf-monoslam_for_3d-reconstruction_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hpp", ".hip")):
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def committed_mfma_busy(kernel_prefix):
    """SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES-normalised duration) of the dominant kernel from the newest
    committed counter pass (profiles/r<k>_pmc_mfma.json, tools/pmc_mfma.py) whose csrc fingerprint matches these sources."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma.json")),
                   key=lambda f: -int(re.search(r"r(\d+)_pmc", os.path.basename(f)).group(1)))
    for path in files:
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        if doc.get("csrc_sha16") != csrc_sha():
            continue
        for name, row in doc.get("kernels", {}).items():
            if name.startswith(kernel_prefix) and "mfma_busy" in row:
                return round(float(row["mfma_busy"]), 4), f"profiles/{os.path.basename(path)} (csrc sha {doc['csrc_sha16']})"
    return None, f"no committed MFMA counter pass matches the kernel sources of this run (csrc sha {csrc_sha()})"


def library_identity(pkg):
    """Which binary ran (an EKF_LIB_PATH left over from an A/B run must not go unnoticed)."""
    import hashlib
    from ekf_monoslam_amd import capi
    path = capi.LIB_PATH
    try:
        sha = hashlib.sha1(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        sha = None
    return {"path": os.path.relpath(path, ROOT) if path.startswith(ROOT) else path, "sha16": sha,
            "overridden_by_EKF_LIB_PATH": bool(os.environ.get("EKF_LIB_PATH")), "csrc_sha16": csrc_sha()}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--features", type=int, default=1000)
    ap.add_argument("--device-warmup-ms", type=float, default=80.0,
                    help="before the W warm-up steps: this many ms of the same step on a SCRATCH map (never the measured one), so that the "
                         "clocks the device holds under this load are reached before the timed region, not inside it (a 20-step window "
                         "behind thousands of tiny map-building launches otherwise sits in the clock ramp); 0 disables; reported in the line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-propagate-pass", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two short rocprofv3 --pmc child passes that measure `roofline.traffic` in this run")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--cpu-full", action="store_true",
                    help="also time ONE frame of the dense formulation on a single thread (minutes at N = 1000; for profiles/)")
    ap.add_argument("--pipeline", type=int, default=-1, help="EKF_OPT_PIPELINE (overlap chain with solve/downdate pieces)")
    ap.add_argument("--exact-fp32", action="store_true",
                    help="EKF_OPT_SPLIT_BF16 = 0: every contraction on v_mfma_f32_32x32x2_f32 (the round 1-4 arithmetic).  The default "
                         "(round 5) runs the covariance downdate of large maps on the bf16 matrix pipe at fp32 accuracy: each "
                         "fp32 operand split exactly into 3 bf16, six of the nine products accumulated in fp32 (dropped: <= 2^-24 |a||b|)")
    ap.add_argument("--no-secondary-exact-fp32", action="store_true",
                    help="skip the short secondary pass that times the same steps with EKF_OPT_SPLIT_BF16 = 0")
    ap.add_argument("--resize-every", type=int, default=0,
                    help="configs[4] cadence: every K frames remove 1 %% of the features and add as many (SURVEY 8d); adds a "
                         "`resize` object (ms per event, compaction GB/s, add cost per feature) from a separate pass")
    return ap.parse_args()


def build_filter(pkg, cfg, n_feat, px0):
    flt = pkg.VSlamFilter(cfg, capacity_features=n_feat, dtype=np.float32)
    flt.setDt(1.0 / 30.0)
    for (u, v) in px0:
        if flt.addFeature((u, v)) != 1:
            raise RuntimeError("synthetic pixel rejected by addFeature")
    flt.synchronize()
    return flt


def segment_frames(n_feat):
    """Frames one map runs before the stream continues on a map started afresh from its current pixels (the work per
    step is the same).  With EVERY feature measured in EVERY frame the fp32 covariance stays positive to rounding for
    as long as it was followed (N = 200: 12000 frames, 400: 6000, 1000: 3000, 2000: 2500, 4000: 1200;
    tools/drift_probe.py, tools/long_run.py, profiles/r2_drift_after_fix.txt); the ring only bounds what was verified."""
    return 3000 if n_feat <= 1000 else 1200


class FilterRing:
    """`frames` frames of the stream on ceil(frames / seg) maps built before the clock starts; frame f runs on
    map f // seg, whose features were initialised from the pixels of frame seg * (f // seg) - 1."""

    def __init__(self, pkg, cfg, n_feat, px0, z, frames, options=()):
        self.seg = segment_frames(n_feat)
        self.filters = []
        nseg = max(1, -(-frames // self.seg))
        if nseg > 64:
            raise SystemExit(f"--steps {frames}: more than 64 map restarts of {self.seg} frames at N = {n_feat}; use fewer steps")
        for s in range(nseg):
            flt = build_filter(pkg, cfg, n_feat, px0 if s == 0 else z[s * self.seg - 1])
            for k, v in options:
                flt.set_option(k, v)
            self.filters.append(flt)

    def at(self, frame):
        return self.filters[min(frame // self.seg, len(self.filters) - 1)]

    def set_option(self, k, v):
        for f in self.filters:
            f.set_option(k, v)

    def profile_reset(self):
        for f in self.filters:
            f.profile_reset()

    def profile(self):
        out = {}
        for f in self.filters:
            for k, (ms, cnt) in f.profile().items():
                a = out.get(k, (0.0, 0))
                out[k] = (a[0] + ms, a[1] + cnt)
        return out

    def profile_work(self):
        out = {}
        for f in self.filters:
            for k, w in f.profile_work().items():
                out[k] = out.get(k, 0.0) + w
        return out

    def synchronize(self):
        for f in self.filters:
            f.synchronize()

    def close(self):
        for f in self.filters:
            f.close()


def run_steps(ring, d_z, d_idx, n_feat, first, count, bytes_per_frame):
    for f in range(first, first + count):
        flt = ring.at(f) if isinstance(ring, FilterRing) else ring
        flt.predict()
        flt.update_device(d_z.data_ptr() + f * bytes_per_frame, d_idx.data_ptr(), n_feat, False)


def _oracle_pair(o, ocfg, n_feat, px0):
    s = o.StructuredFilter(ocfg, np.float32)
    s.dT = 1.0 / 30.0
    for (u, v) in px0:
        assert s.add_feature(u, v) == 1
    d = o.DenseFilter(ocfg, np.float32)
    d.dT = s.dT
    d.mu, d.Sigma = s.mu.copy(), s.Sigma.copy()
    d.features = [o.Feature(position_in_state=f.position_in_state) for f in s.features]
    return s, d


def cpu_baseline(cfg_name, n_feat, px0, zs, threads, full=False):
    """Predict + update frames of the same workload on the host cores (a bounded sample, ~20 s in all):
      dense   the reference's formulation (oracle.DenseFilter: F Sigma F^T, H Sigma H^T, Sigma H^T S^-1, (I - K H) Sigma,
              Qc Sigma Qc^T as dense n^3 products, vR.cpp:457-477, 598, 1268-1280, 1641), sgemm / inverse from OpenBLAS
      structured  the same update exploiting the identity blocks (O(n^2 m))
    on all cores, and on ONE thread (the reference builds without OpenMP, mono-slam/CMakeLists.txt:3: single-thread SSE4
    Eigen): the structured port is timed on one thread; one dense frame on one thread is ~3.5e12 flop at N = 1000
    (over a minute), so by default it is ESTIMATED from the measured single-thread sgemm rate (flagged as such) and only
    measured with --cpu-full.  /usr/include/eigen3 is probed: with Eigen on the box the reference's own expressions could
    be compiled; without it (the case so far) this port is the baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ekf_oracle as o
    from threadpoolctl import threadpool_info, threadpool_limits
    ocfg = o.Config.kinect()
    idx = list(range(n_feat))
    out = {"eigen3_on_box": os.path.isdir("/usr/include/eigen3") or os.path.isdir("/usr/local/include/eigen3")}

    def timed(filt, max_frames, budget_s):
        ts = []
        end = time.perf_counter() + budget_s
        for k in range(min(max_frames, len(zs))):
            t0 = time.perf_counter()
            filt.predict()
            filt.update(zs[k].reshape(-1), idx)
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() > end:
                break
        return float(np.median(ts)), len(ts)

    with threadpool_limits(limits=threads):
        used = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
        s, d = _oracle_pair(o, ocfg, n_feat, px0)
        t_dense, n_dense = timed(d, 20, 8.0)
        t_struct, n_struct = timed(s, 20, 3.0)
    out.update({"dense_s": t_dense, "dense_frames": n_dense, "structured_s": t_struct, "threads": used})
    with threadpool_limits(limits=1):
        s1, d1 = _oracle_pair(o, ocfg, n_feat, px0)
        t_struct1, n1 = timed(s1, 5, 6.0)
        # single-thread sgemm rate on a 2048^3 product (the dense frame is n^3 sgemm work to > 95 %)
        a = np.random.default_rng(0).standard_normal((2048, 2048)).astype(np.float32)
        a @ a
        t0 = time.perf_counter()
        a @ a
        rate = 2 * 2048.0 ** 3 / (time.perf_counter() - t0)
        n = d1.n
        m = 2 * n_feat
        # predict 2 x 2n^3, S 2 m n^2 + 2 m^2 n, K 2 n^2 m + 2 n m^2 + ~2 m^3 (inverse), (I - K H) Sigma 2 n^2 m + 2 n^3, normalise 2 x 2n^3
        dense_flop = 4.0 * n ** 3 + 2.0 * m * n * n + 2.0 * m * m * n + 2.0 * n * n * m + 2.0 * n * m * m + 2.0 * m ** 3 \
            + 2.0 * n * n * m + 2.0 * n ** 3 + 4.0 * n ** 3
        out.update({"structured_1thread_s": t_struct1, "sgemm_1thread_gflops": rate / 1e9,
                    "dense_1thread_s_estimate": dense_flop / rate, "dense_flop_per_frame": dense_flop})
        if full:
            t_dense1, _ = timed(d1, 1, 1.0)
            out["dense_1thread_s_measured"] = t_dense1
    return out


def resize_pass(pkg, cfg, n_feat, px0, z, args):
    """configs[4] cadence on ONE GPU: every `--resize-every` frames 1 % of the features are removed (uniform
    indices, seed 1236, one call: the library removes in descending order as vR.cpp:1296 does) and as many are added
    (vR.cpp:309-371, one ekf_add_feature each); the new features are never measured (no stream behind them).  Wall time of
    the two halves of an event between synchronisations, and the kernels behind them from the library's HIP events:
    k_compact_transform (vR.cpp:373-421: Sigma read once and written once into the second buffer, 2 n^2 s algorithmic
    bytes) and k_add_prepare + k_add_border (vR.cpp:309-371: an O(n) border per feature)."""
    every = args.resize_every
    frames = min(len(z), args.warmup + args.steps)
    flt = build_filter(pkg, cfg, n_feat, px0)
    rng = np.random.default_rng(1236)
    sid = np.arange(n_feat)
    t_rm, t_add, n_rm = [], [], []
    kern = {}
    t_steps = 0.0
    for k in range(frames):
        t0 = time.perf_counter()
        flt.predict()
        h, vis, rem, _ = flt.predictions()
        sel = np.nonzero(vis.astype(bool) & (sid >= 0))[0].astype(np.int32)
        flt.update(z[k][sid[sel]].reshape(-1), sel)
        if (k + 1) % every == 0:
            flt.synchronize()
            t_steps += time.perf_counter() - t0
            N = flt.numOfFeatures()
            n_state = flt.stateDim()
            drop = sorted(rng.choice(N, size=max(1, N // 100), replace=False).tolist())
            flt.set_option(2, 2)
            flt.profile_reset()
            t1 = time.perf_counter()
            flt.removeFeatures(drop)
            flt.synchronize()
            t2 = time.perf_counter()
            for _ in range(len(drop)):
                assert flt.addFeature((float(rng.uniform(20, 300)), float(rng.uniform(20, 220)))) == 1
            flt.synchronize()
            t3 = time.perf_counter()
            for name, (ms, cnt) in flt.profile().items():
                a = kern.setdefault(name, [0.0, 0])
                a[0] += ms
                a[1] += cnt
            flt.set_option(2, 0)
            sid = np.concatenate([np.delete(sid, drop), -np.ones(len(drop), np.int64)])
            t_rm.append((t2 - t1, n_state))
            t_add.append(t3 - t2)
            n_rm.append(len(drop))
        else:
            t_steps += 0.0
    flt.synchronize()
    if not t_rm:
        flt.close()
        return {"note": f"--resize-every {every}: no event inside {frames} frames"}
    n_state = int(np.median([ns for _, ns in t_rm]))
    cms, ccnt = kern.get("compact_transform", (0.0, 0))
    ams, acnt = kern.get("add_feature", (0.0, 0))
    out = {"every_frames": every, "events": len(t_rm), "features_removed_and_added_per_event": int(np.median(n_rm)),
           "state_dim": n_state,
           "remove_ms_per_event": round(1e3 * float(np.median([t for t, _ in t_rm])), 3),
           "add_ms_per_event": round(1e3 * float(np.median(t_add)), 3),
           "add_ms_per_feature": round(1e3 * float(np.median(t_add)) / max(1, int(np.median(n_rm))), 4),
           "basis": "wall time between synchronisations, per event (a removal call of 1 % of the features, then one "
                    "ekf_add_feature per new feature); kernels from EKF_OPT_PROFILE = 2 events during the events only"}
    if ccnt:
        t_k = cms / ccnt * 1e-3
        nbytes = 2.0 * n_state * n_state * 4
        out["compact_transform"] = {"kernel": "k_compact_transform (vR.cpp:373-421, 741-772)", "bound": "hbm", "launches": ccnt,
                                    "avg_launch_ms": round(t_k * 1e3, 4), "algorithmic_bytes_per_launch": nbytes,
                                    "achieved": round(nbytes / t_k / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": round(nbytes / t_k / 1e9 / PEAK_HBM_GBS, 4)}
    if acnt:
        out["add_feature_kernels"] = {"kernels": "k_add_prepare + k_add_border (vR.cpp:309-371)", "launches": acnt,
                                      "avg_launch_ms": round(ams / acnt, 4)}
    flt.close()
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks as fresh child processes through
    torch.distributed.run, exactly as the driver does, forward rank 0's JSON line and exit with the child's
    code.  Nothing in THIS process has touched the GPU yet (no os.exec from a process that has: the children are
    ordinary subprocesses)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def under_profiler():
    """True when this process was started by rocprofv3 (its tool library is preloaded and has initialised the GPU: a
    child must not be exec'd from here)."""
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
        return True
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)


def live_pmc_traffic(args):
    """HBM bytes per launch of every kernel of THIS workload, measured now: two child runs of this script (5 steps)
    under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` (separate passes, counters
    only; MI355X_MICROARCH.md, HBM section: KB per dispatch, the read counter doubled on gfx950).  Called before this
    process touches the GPU.  Returns ({kernel: bytes per launch}, note) or (None, reason)."""
    import csv
    import glob
    import re
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    out = {}
    work = tempfile.mkdtemp(prefix="ekf_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", EKF_BENCH_CHILD="1")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__), "--features", str(args.features), "--steps", "5", "--warmup", "2",
                   "--no-cpu-baseline", "--no-live-traffic", "--pipeline", str(args.pipeline)]
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                rc = proc.wait(timeout=180)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)          # the process group of the child, nothing else
                proc.wait()
                return None, f"rocprofv3 --pmc {counter} child pass timed out"
            files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
            if rc != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} child pass failed (rc {rc})"
            acc = {}
            for r in csv.DictReader(open(files[-1])):
                if r.get("Counter_Name") != counter:
                    continue
                k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ekf::", "").replace("ekf::", "")
                a = acc.setdefault(k, [0.0, 0])
                a[0] += float(r["Counter_Value"])
                a[1] += 1
            for k, (tot, cnt) in acc.items():
                out.setdefault(k, {})[counter] = tot / cnt
    finally:
        shutil.rmtree(work, ignore_errors=True)
    traffic = {k: {"hbm_bytes_per_launch": (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0} for k, v in out.items()}
    return traffic, ("measured in this run: two child passes of this script (5 steps) under rocprofv3 --pmc FETCH_SIZE / "
                     "--pmc WRITE_SIZE with --kernel-trace only; KB per dispatch, FETCH_SIZE x2 (gfx950)")


LIVE_PMC = (None, None)


def main():
    global LIVE_PMC
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={env_world} of the launcher")
    # (sizes above the headline workload are left to the committed passes: the profiler's counter pass itself crashed
    # on the N = 4000 run, rc -11, and nothing here should depend on it)
    if (env_world is None and args.gpus == 1 and not args.no_live_traffic and not os.environ.get("EKF_BENCH_CHILD")
            and args.features <= 1000 and not under_profiler()):
        LIVE_PMC = live_pmc_traffic(args)                # before anything here initialises the GPU
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    pkg = load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("EKF_BENCH_BACKEND", "nccl")      # "gloo": rehearsal with ranks sharing one GPU
        ndev = torch.cuda.device_count()
        local_rank = local_rank % max(ndev, 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)

    n_feat = args.features
    cfg = pkg.kinect_config()
    frames = args.warmup + args.steps
    from ekf_monoslam_amd import synthetic
    px0, z = synthetic.measurement_stream(cfg, n_feat, frames, sigma_px=SIGMA_Z_PX)
    if world > 1:
        from ekf_monoslam_amd import sharded
        result = sharded.bench(pkg, cfg, n_feat, px0, z, args, rank, world, dev)
    else:
        result = bench_single(pkg, cfg, n_feat, px0, z, args, dev, torch)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def bench_single(pkg, cfg, n_feat, px0, z, args, dev, torch):
    frames = args.warmup + args.steps
    opts = [(3, args.pipeline)] + ([(4, 0)] if args.exact_fp32 else [])
    flt = FilterRing(pkg, cfg, n_feat, px0, z, frames, opts)
    n = flt.filters[0].stateDim()
    d_z = torch.from_numpy(z.reshape(z.shape[0], -1)).to(dev).contiguous()
    d_idx = torch.arange(n_feat, dtype=torch.int32, device=dev)
    bpf = 2 * n_feat * 4
    torch.cuda.synchronize()

    # device warm-up on a scratch map (see --device-warmup-ms): the measured maps are not touched, the W warm-up steps and the
    # K timed steps follow unchanged
    device_warmup = {"requested_ms": args.device_warmup_ms, "steps_on_scratch_map": 0, "ms": 0.0}
    if args.device_warmup_ms > 0:
        scratch = build_filter(pkg, cfg, n_feat, px0)
        for k, v in opts:
            scratch.set_option(k, v)
        tw = time.perf_counter()
        nw = 0
        while (time.perf_counter() - tw) * 1e3 < args.device_warmup_ms and nw < segment_frames(n_feat):
            f_ = nw % z.shape[0]
            scratch.predict()
            scratch.update_device(d_z.data_ptr() + f_ * bpf, d_idx.data_ptr(), n_feat, False)
            nw += 1
            if nw % 8 == 0:
                scratch.synchronize()
        scratch.synchronize()
        device_warmup.update(steps_on_scratch_map=nw, ms=round((time.perf_counter() - tw) * 1e3, 2))
        scratch.close()
        del scratch

    run_steps(flt, d_z, d_idx, n_feat, 0, args.warmup, bpf)
    flt.synchronize()
    torch.cuda.synchronize()
    # EKF_OPT_PROFILE: HIP events around the dominant kernel only, and only in every 8th frame of the timed region
    # (an event pair costs ~6 us of queue time; runs of fewer than 16 steps time every frame)
    sample = 8 if args.steps >= 16 else 1
    flt.set_option(2, 3 if sample == 8 else 1)
    flt.profile_reset()
    t0 = time.perf_counter()
    run_steps(flt, d_z, d_idx, n_feat, args.warmup, args.steps, bpf)
    flt.synchronize()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    prof = flt.profile()
    work = flt.profile_work()
    flt.set_option(2, 0)
    elapsed = t1 - t0
    ms_per_step = 1e3 * elapsed / args.steps

    # sanity of the run: finite state, unit quaternion, every feature still in view
    last = flt.at(frames - 1)
    mu = last.getFullState()
    last.predict()
    h, vis, rem, S2 = last.predictions()
    sane = bool(np.all(np.isfinite(mu)) and abs(np.linalg.norm(mu[3:7]) - 1) < 1e-4 and int(vis.sum()) >= int(0.98 * n_feat))

    m = 2 * n_feat
    syrk_ms, syrk_cnt = prof.get("downdate_syrk", (0.0, 0))
    roofline = None
    pieces = 1
    # which arithmetic the downdate ran in: the library switches to the bf16x6 kernel for maps of >= 23 tile rows
    # (round 6: read from the library's launch counters -- ekf_launch_count -- instead of re-deriving its selection rule)
    launches = {}
    for f_ in flt.filters:
        for k_, v_ in f_.launch_counts().items():
            launches[k_] = launches.get(k_, 0) + v_
    split_used = launches.get("downdate_bf16x6", 0) > 0
    if split_used and (launches.get("downdate_f32", 0) + launches.get("downdate_f32_fused_wu", 0) + launches.get("downdate_f32_half_tail", 0)) > 0:
        raise SystemExit("bench: the downdate ran on both arithmetics in one run (a per-chunk fall-back): the line would misreport dtype / roofline")
    if syrk_cnt:
        # the timed launches and their algorithmic flop come from the library (ekf_profile_read / ekf_profile_work):
        # n^2 x the real columns of every downdate launch (symmetric half).  Exact-fp32 path with the default
        # pipeline: the first launches also carry the right-looking update of the innovation ROW of their chunk
        # (2 x 1 x (m - c1) x its columns; the whole W, 2 (n + 1) (m - c1) x its columns, under EKF_OPT_W_RECOMPUTE = 0).
        t_k = syrk_ms / syrk_cnt * 1e-3
        # launches per step: the timed launches / the steps that were timed (every `sample`-th of the K steps)
        pieces = max(1, round(syrk_cnt / max(1, -(-args.steps // sample))))
        flop = work.get("downdate_syrk", 0.0) / syrk_cnt
        ach = flop / t_k / 1e12
        note = ("%d launches per step (column chunks of V; all but the last on 224 of 256 CUs beside the serial Cholesky "
                "chain, EKF_OPT_PIPELINE); --pipeline 0 runs one launch" % pieces) if pieces > 1 or args.pipeline != 0 \
            else "one launch per step"
        if split_used:
            # the kernel executes SIX bf16 products per algorithmic fp32 product: its MFMA roofline is the dense bf16 peak
            # over 6.  `achieved` stays the ALGORITHMIC fp32 flop over the measured duration.
            peak6 = PEAK_BF16_MFMA_TF / 6.0
            roofline = {"kernel": "downdate_syrk (k_syrk_bf16x6: v_mfma_f32_32x32x16_bf16 on 3 x bf16 split operands, six "
                                  "products per fp32 product, fp32 accumulate; operands by LDS-DMA from the plane image of V_g)",
                        "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak6, 1),
                        "unit": "TFLOP/s (algorithmic fp32 flop)", "frac": round(ach / peak6, 4), "traffic": None,
                        "peak_basis": "dense bf16 MFMA peak %.0f TFLOP/s / 6 products per fp32 product" % PEAK_BF16_MFMA_TF,
                        "executed_bf16_tflops": round(6.0 * ach, 1),
                        "vs_f32_mfma_peak": round(ach / PEAK_F32_MFMA_TF, 4),
                        "avg_launch_ms": round(t_k * 1e3, 4), "launches": syrk_cnt, "timed_every_nth_step": sample,
                        "algorithmic_flop_per_launch": flop, "launches_per_step": pieces, "note": note}
        else:
            roofline = {"kernel": "downdate_syrk (k_gemm_mfma<DOWNDATE>, f32 MFMA 32x32x2)", "bound": "mfma",
                        "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_F32_MFMA_TF, 4), "traffic": None,
                        "avg_launch_ms": round(t_k * 1e3, 4), "launches": syrk_cnt, "timed_every_nth_step": sample,
                        "algorithmic_flop_per_launch": flop, "launches_per_step": pieces, "note": note}

    # the whole step against the f32 MFMA peak: algorithmic flop of the formulation THIS run executed, from the library's
    # own column chunks (ekf_get_chunk_plan): symmetric-half downdate n^2 m; V_g = W_g Z_gg per chunk, n w_g^2 (Z_gg upper
    # triangular); right-looking W update 2 n w_g (m - c1_g) per chunk, or -- EKF_OPT_W_RECOMPUTE, the default -- the
    # re-evaluation of the later chunks' W from the downdated Sigma, 26 n w; Cholesky of S m^3 / 3 and the chunk
    # inverses sum w_g^3 / 3; W and S in front, 26 n m + 26 m^2.  (Rounds 1-3 ran 99 GFLOP per step at N = 1000; the
    # sequential form runs 86.)
    block, ends, wrec = last.chunkPlan()
    widths = [(e - (ends[g - 1] if g else 0)) * block for g, e in enumerate(ends)] or [m]
    c1s = [e * block for e in ends] or [m]
    solve_flop = sum(float(n) * w * w for w in widths)
    wupd_flop = 0.0 if wrec else sum(2.0 * n * w * max(0, m - c1) for w, c1 in zip(widths, c1s))
    reval_flop = sum(26.0 * n * w for w in widths[1:]) if wrec else 0.0
    chol_flop = float(m) ** 3 / 3.0 + sum(float(w) ** 3 / 3.0 for w in widths)
    step_flop = float(n) * n * m + solve_flop + wupd_flop + reval_flop + chol_flop + 26.0 * n * m + 26.0 * m * m
    ach_step = step_flop / (ms_per_step * 1e-3) / 1e12
    # (round 6: no fraction any more -- the step mixes the bf16x6 downdate, f32-MFMA solves and a latency-bound chain, and a
    # fraction of any ONE peak says nothing; the achieved figure and the flop breakdown are kept for continuity)
    roofline_step = {"bound": "mixed: bf16x6 downdate (matrix pipe), f32-MFMA solves, latency-bound Cholesky chain" if n_feat >= 600 else "latency",
                     "achieved": round(ach_step, 2), "peak": None, "unit": "TFLOP/s (algorithmic fp32 flop of the whole step)",
                     "frac": None, "traffic": None,
                     "algorithmic_flop_per_step": step_flop,
                     "flop_breakdown": {"downdate": float(n) * n * m, "solve": solve_flop, "w_update": wupd_flop,
                                        "w_reevaluation": reval_flop, "cholesky_and_chunk_inverses": chol_flop},
                     "chunk_ends_block_steps": ends, "block": block, "w_recompute": wrec,
                     "flop_of_the_round_1_3_formulation": float(n) * n * m + float(n) * m * m / 2.0 * 1.1 + chol_flop
                     + sum(2.0 * n * w * max(0, m - c1) for w, c1 in zip(widths, c1s)) + 26.0 * n * m + 26.0 * m * m,
                     "basis": "whole step: algorithmic flop of the formulation executed (the library's column chunks) / ms_per_step"}

    # HBM traffic per launch: measured by two rocprofv3 --pmc child passes of this run (live_pmc_traffic); when those are
    # not available (child of a profiler, no rocprofv3, --no-live-traffic) from the committed PMC passes, which belong to
    # the kernel sources they were collected on: the file carries a fingerprint of csrc/, and `traffic` is then reported
    # only while it matches the sources of THIS run (else null, with the reason).
    pmc, pmc_note = {}, None
    if LIVE_PMC[0]:
        pmc, pmc_note = LIVE_PMC
    elif LIVE_PMC[1]:
        pmc_note = LIVE_PMC[1]
    if not pmc and n_feat == 1000:
        # the newest committed pass whose csrc fingerprint matches the sources of this run (profiles/r<k>_pmc_traffic.json)
        import glob
        import re
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")),
                       key=lambda f: -int(re.search(r"r(\d+)_pmc", os.path.basename(f)).group(1)))
        seen = []
        for pmc_path in files:
            doc = json.load(open(pmc_path))
            seen.append(f"{os.path.basename(pmc_path)}:{doc.get('csrc_sha16')}")
            if doc.get("csrc_sha16") == csrc_sha():
                pmc = doc["kernels"]
                pmc_note = (f"profiles/{os.path.basename(pmc_path)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                            f"FETCH_SIZE x2), collected on these kernel sources (csrc sha {doc['csrc_sha16']})")
                if LIVE_PMC[1]:
                    pmc_note = f"live pass unavailable ({LIVE_PMC[1]}); " + pmc_note
                break
        else:
            pmc_note = ((f"live pass unavailable ({LIVE_PMC[1]}); " if LIVE_PMC[1] else "") +
                        f"no committed PMC pass matches the kernel sources of this run (csrc sha {csrc_sha()}; have "
                        f"{', '.join(seen) or 'none'}): traffic not reported")
    dd_key = next((k for k in pmc if k.startswith("k_syrk_bf16x6" if split_used else "k_gemm_mfma<2, false")), None)
    if roofline:
        roofline["traffic"] = pmc[dd_key]["hbm_bytes_per_launch"] if dd_key else None
        roofline["traffic_source"] = pmc_note
        roofline["mfma_busy"], roofline["mfma_busy_source"] = committed_mfma_busy("k_syrk_bf16x6" if split_used else "k_gemm_mfma<2, false")
    result = {
        "metric": "EKF updates/sec at N features (state dim 14+6N)",
        "value": round(args.steps / elapsed, 2), "unit": "updates/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32" if not split_used else
                 "f32 (state, covariance, every accumulation and every other contraction in fp32; the products of the covariance "
                 "downdate as 3 x bf16 splits of the fp32 operands, six bf16 products per fp32 product: the dropped terms are "
                 "<= 2^-24 |a||b| in the worst case (2^-28 on average) -- the size of the fp32 product's own rounding; EKF_OPT_SPLIT_BF16 = 0 / --exact-fp32 runs v_mfma_f32_32x32x2_f32)",
        "data": "synthetic",
        "config": {"workload": f"N={n_feat} inverse-depth features, n={n}, M=N measured per frame, "
                               f"fp32, 1xMI355X ({ {200: 'BASELINE configs[1]', 1000: 'BASELINE configs[2]', 4000: 'BASELINE configs[4] size on one GPU'}.get(n_feat, 'custom size') })",
                   "features": n_feat, "state_dim": n, "measured_per_frame": n_feat,
                   "camera": "conf_kinect.cfg/scale2", "dT": 1.0 / 30.0,
                   "frames_per_map": flt.seg, "maps": len(flt.filters)},
        "device_warmup": device_warmup,
        "library": library_identity(pkg),
        "downdate_launch_kinds": {k_: v_ for k_, v_ in launches.items() if k_.startswith(("downdate", "row_", "chain"))},
        "run_sane": sane, "features_visible_at_end": int(vis.sum()), "features_rho_nonpositive_at_end": int(rem.sum()),
        "roofline": roofline,
        "roofline_step": roofline_step,
        "kernel_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()},
    }

    flt.close()                             # the secondary passes build their own maps

    if not args.no_propagate_pass:
        # secondary figure: the step as a drop-in caller drives it (vR.cpp:868-875: update() reads every patch's h and
        # the 2x2 St blocks that predict() left, and the matcher's z arrives from the host): per frame
        # ekf_predict -> ekf_get_predictions (h, flags, 2x2 St blocks: D2H + one synchronisation) -> ekf_update with z and
        # the index list from HOST memory (H2D).  Not `value`: the headline keeps inputs resident (bench contract).
        flt5 = FilterRing(pkg, cfg, n_feat, px0, z, frames)
        idx_h = np.arange(n_feat, dtype=np.int32)

        def run_dropin(first, count):
            for f in range(first, first + count):
                f5 = flt5.at(f)
                f5.predict()
                f5.predictions()
                f5.update(z[f].reshape(-1), idx_h)
        run_dropin(0, args.warmup)
        flt5.synchronize()
        t0 = time.perf_counter()
        run_dropin(args.warmup, args.steps)
        flt5.synchronize()
        t1 = time.perf_counter()
        result["secondary_dropin_step"] = {
            "what": "predict + ekf_get_predictions (h, flags, 2x2 St blocks to the host) + ekf_update from host z / indices",
            "value": round(args.steps / (t1 - t0), 2), "unit": "updates/s", "ms_per_step": round(1e3 * (t1 - t0) / args.steps, 4)}
        flt5.close()

        # secondary workload: M = 32 measured features per frame, the reference's real
        # operating point (conf_sim.cfg:24-25) -- the dense contractions shrink to rank 64 and the step
        # becomes HBM-bound (W pass + rank-64 downdate stream Sigma)
        m32 = min(32, n_feat)
        flt3 = FilterRing(pkg, cfg, n_feat, px0, z, frames)
        sel = torch.arange(m32, dtype=torch.int32, device=dev)
        d_z32 = d_z[:, :2 * m32].contiguous()

        def run32(first, count):
            for f in range(first, first + count):
                f3 = flt3.at(f)
                f3.predict()
                f3.update_device(d_z32.data_ptr() + f * 2 * m32 * 4, sel.data_ptr(), m32, False)
        run32(0, args.warmup)
        flt3.synchronize()
        t0 = time.perf_counter()
        run32(args.warmup, args.steps)
        flt3.synchronize()
        t1 = time.perf_counter()
        t32 = (t1 - t0) / args.steps
        b32 = 3.0 * n * n * 4 + 4.0 * n * 2 * m32 * 4        # Sigma read for W, read + written by the downdate; W, V
        result["secondary_M32"] = {"measured_per_frame": m32, "value": round(args.steps / (t1 - t0), 2),
                                   "unit": "updates/s", "ms_per_step": round(1e3 * t32, 4),
                                   "roofline": {"bound": "hbm", "achieved": round(b32 / t32 / 1e9, 1), "peak": PEAK_HBM_GBS,
                                                "unit": "GB/s", "frac": round(b32 / t32 / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                                                "basis": "whole step: algorithmic bytes 3 n^2 s + 4 n 2M s (rank-64 update: the "
                                                         "contractions are below the MFMA / HBM crossover, SURVEY 8d) / ms_per_step",
                                                "algorithmic_bytes_per_step": b32}}
        flt3.close()

        if n_feat < 600:
            # small maps (configs[1], N = 200) are LATENCY-bound: one column chunk, a serial chain of a few block steps and
            # ~20 launches per step; say so with numbers next to the MFMA roofline of the (tiny) downdate
            flt6 = FilterRing(pkg, cfg, n_feat, px0, z, min(frames, 60))
            flt6.set_option(2, 2)                            # HIP events around every kernel (adds launch gaps: shares only)
            flt6.profile_reset()
            k6 = min(40, frames)
            run_steps(flt6, d_z, d_idx, n_feat, 0, k6, bpf)
            flt6.synchronize()
            p6 = flt6.profile()
            flt6.set_option(2, 0)
            per = {kname: (ms / k6, cnt / k6) for kname, (ms, cnt) in p6.items() if cnt}
            chain = sum(per.get(kname, (0.0, 0))[0] for kname in ("chol_diag", "chol_panel", "chol_trailing"))
            bytes_step = 3.0 * n * n * 4 + 4.0 * n * 2 * n_feat * 4 + 2.0 * (2 * n_feat) ** 2 * 4
            result["roofline_step"] = {
                "bound": "latency", "achieved": round(bytes_step / (ms_per_step * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": round(bytes_step / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                "basis": "whole step: algorithmic bytes 3 n^2 s + 4 n m s + 2 m^2 s / ms_per_step -- far from either roofline: "
                         "the step is the sum of its launch latencies",
                "launches_per_step": round(sum(c for _, c in per.values()), 1),
                "launches_per_step_note": "counted under the per-kernel profile, which runs one launch per kernel; the timed run "
                                          "(EKF_OPT_FUSED_LAUNCHES, default) needs " +
                                          ("5: predict, W, S, diagonal factor, fused update (2M + 3 <= 128: one diagonal block)"
                                           if 2 * n_feat + 3 <= 128 else "2 fewer (ekf_predict is one launch)"),
                "chain_block_steps": int(round(per.get("chol_diag", (0.0, 0))[1])),
                "chain_kernel_ms_per_step": round(chain, 4),
                "chain_share_of_kernel_time": round(chain / max(sum(v for v, _ in per.values()), 1e-9), 3),
                "kernel_ms_per_step": {kname: round(v, 4) for kname, (v, _) in sorted(per.items(), key=lambda kv: -kv[1][0])[:8]}}
            flt6.close()

        if split_used and not args.no_secondary_exact_fp32:
            # the same steps with every contraction on the f32 matrix instruction (EKF_OPT_SPLIT_BF16 = 0: the arithmetic of
            # rounds 1-4), so that the line carries both figures
            flt4 = FilterRing(pkg, cfg, n_feat, px0, z, frames, [(3, args.pipeline), (4, 0)])
            run_steps(flt4, d_z, d_idx, n_feat, 0, args.warmup, bpf)
            flt4.synchronize()
            t0 = time.perf_counter()
            run_steps(flt4, d_z, d_idx, n_feat, args.warmup, args.steps, bpf)
            flt4.synchronize()
            t1 = time.perf_counter()
            mu4 = flt4.at(frames - 1).getFullState()
            result["secondary_exact_fp32"] = {
                "option": "EKF_OPT_SPLIT_BF16 = 0 (every contraction v_mfma_f32_32x32x2_f32)", "value": round(args.steps / (t1 - t0), 2),
                "unit": "updates/s", "ms_per_step": round(1e3 * (t1 - t0) / args.steps, 4),
                "max_abs_mu_difference_to_the_default_path": float(np.abs(mu4 - mu).max()),
                "run_sane": bool(np.all(np.isfinite(mu4)) and abs(np.linalg.norm(mu4[3:7]) - 1) < 1e-4)}
            flt4.close()

        # second pass, same steps, streaming P-propagate: HBM GB/s of P <- F P F^T + Q
        flt2 = FilterRing(pkg, cfg, n_feat, px0, z, frames, [(0, 1)])
        run_steps(flt2, d_z, d_idx, n_feat, 0, args.warmup, bpf)
        flt2.synchronize()
        flt2.set_option(2, 1)
        flt2.profile_reset()
        t0 = time.perf_counter()
        run_steps(flt2, d_z, d_idx, n_feat, args.warmup, args.steps, bpf)
        flt2.synchronize()
        t1 = time.perf_counter()
        p2 = flt2.profile()
        ms, cnt = p2.get("propagate_streaming", (0.0, 0))
        if cnt:
            t_k = ms / cnt * 1e-3
            nbytes = 2.0 * n * n * 4
            gbs = nbytes / t_k / 1e9
            result["p_propagate"] = {"kernel": "k_propagate_streaming", "bound": "hbm",
                                     "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                     "frac": round(gbs / PEAK_HBM_GBS, 4),
                                     "traffic": pmc.get("k_propagate_streaming<float>", {}).get("hbm_bytes_per_launch"),
                                     "avg_launch_ms": round(t_k * 1e3, 4),
                                     "algorithmic_bytes_per_launch": nbytes,
                                     "updates_per_s_with_streaming_propagate": round(args.steps / (t1 - t0), 2)}
        # the default in-place strip kernel, timed in a short pass of its own (every-kernel events add launch gaps)
        flt2.set_option(0, 0)
        flt2.set_option(2, 2)
        flt2.profile_reset()
        for _ in range(20):
            flt2.filters[0].predict()
        flt2.synchronize()
        ms, cnt = flt2.profile().get("propagate_strips", (0.0, 0))
        flt2.set_option(2, 0)
        if cnt:
            result["p_propagate_in_place"] = {"kernel": "k_strip_congruence<13>", "avg_launch_ms": round(ms / cnt, 4),
                                              "algorithmic_bytes_per_launch": 4.0 * 13 * n * 4}
        flt2.close()

        if n_feat >= 600:
            # (last of the GPU passes: its ~2000 tiny launches leave the clocks low for whatever follows)
            # the reference's own operating point as a map (conf_sim.cfg:20-25: <= 35 features, all of them measured):
            # N = 32, n = 205, 2 M + 3 <= 128 -- one diagonal block, so the update after the diagonal factor is ONE launch
            # (k_update_oneblock_small) and the step is 5 launches: launch-latency-bound, reported as such
            from ekf_monoslam_amd import synthetic
            n32 = 32
            fr32 = 400
            px32, z32 = synthetic.measurement_stream(cfg, n32, fr32, sigma_px=SIGMA_Z_PX)
            flt7 = FilterRing(pkg, cfg, n32, px32, z32, fr32)
            d_z7 = torch.from_numpy(z32.reshape(fr32, -1)).to(dev).contiguous()
            d_i7 = torch.arange(n32, dtype=torch.int32, device=dev)
            bpf7 = 2 * n32 * 4
            run_steps(flt7, d_z7, d_i7, n32, 0, 50, bpf7)
            flt7.synchronize()
            t0 = time.perf_counter()
            run_steps(flt7, d_z7, d_i7, n32, 50, fr32 - 50, bpf7)
            flt7.synchronize()
            t1 = time.perf_counter()
            t7 = (t1 - t0) / (fr32 - 50)
            mu7 = flt7.at(fr32 - 1).getFullState()
            result["secondary_N32_map"] = {
                "what": "a 32-feature map (n = 205), every feature measured in every frame: the reference's operating point",
                "value": round(1.0 / t7, 1), "unit": "updates/s", "ms_per_step": round(1e3 * t7, 4), "launches_per_step": 5,
                "roofline": {"bound": "latency", "basis": "5 dependent launches of 5-10 us each (predict, W, S, diagonal factor, "
                             "fused update); 3 n^2 s = 0.5 MB of traffic per step"},
                "run_sane": bool(np.all(np.isfinite(mu7)) and abs(np.linalg.norm(mu7[3:7]) - 1) < 1e-4)}
            flt7.close()

    if args.resize_every > 0:
        result["resize"] = resize_pass(pkg, cfg, n_feat, px0, z, args)

    if not args.no_cpu_baseline:
        threads = args.cpu_threads or len(os.sched_getaffinity(0))
        cb = cpu_baseline("kinect", n_feat, px0, z, threads, args.cpu_full)
        d1 = cb.get("dense_1thread_s_measured")
        result["cpu_baseline"] = {
            "value": round(1.0 / cb["dense_s"], 4), "unit": "updates/s", "cores": cb["threads"], "kind": "port",
            "sample": f"median of {cb['dense_frames']} frame(s) (predict+update, ~8 s of host time) of the same N={n_feat}, M=N "
                      "workload in the reference's dense n^3 formulation, numpy/OpenBLAS sgemm, every host core",
            "seconds_per_update": round(cb["dense_s"], 3),
            "structured_port_updates_per_s": round(1.0 / cb["structured_s"], 3),
            "one_thread": {
                "why": "the reference builds without OpenMP (mono-slam/CMakeLists.txt:3): its Eigen path is single-threaded",
                "structured_port_updates_per_s": round(1.0 / cb["structured_1thread_s"], 4),
                "dense_port_updates_per_s": round(1.0 / (d1 or cb["dense_1thread_s_estimate"]), 5),
                "dense_port_is": "measured (1 frame)" if d1 else
                                 f"ESTIMATED: {cb['dense_flop_per_frame']:.3g} flop per frame / the measured single-thread sgemm rate "
                                 f"({cb['sgemm_1thread_gflops']:.1f} GFLOP/s); --cpu-full measures it",
            },
            "eigen3_on_box": cb["eigen3_on_box"],
            "eigen_note": "no /usr/include/eigen3 on the box: the reference's own Eigen expressions cannot be compiled here"
                          if not cb["eigen3_on_box"] else "Eigen headers present on the box",
        }
    return result


if __name__ == "__main__":
    main()
