#!/bin/bash
# Round-6 evidence (ROUND=rN overrides the file prefix) on the GPU box, in stages that each fit one gpurun call (<= 20 min):
#   tools/collect_profiles.sh A   bench lines, rocprofv3 kernel stats + timelines, PMC traffic, MFMA counters, resize cadence
#   tools/collect_profiles.sh B   micro-benchmarks, sharded world 1 over RCCL, gloo rehearsals, knob A/B, accuracy of the knobs
#   tools/collect_profiles.sh C   N = 4000 against the fp64 oracle (every entry)
#   tools/collect_profiles.sh D   long all-measured runs (positivity)
#   tools/collect_profiles.sh E   CPU baseline with the measured single-thread dense frame
# outputs under gpurun_out/r6/ (copy the summaries into profiles/)
set -e
R="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
P="${ROUND:-r6}"
O="$R/gpurun_out/$P"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
STAGE="${1:-A}"
if [ "$STAGE" = A ]; then
echo "bench N=1000"; python3 "$R/bench.py" > "$O/${P}_bench.json" 2> "$O/bench.err"
echo "bench with the driver's arguments"; python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 > "$O/${P}_bench_driver_args.json" 2>> "$O/bench.err"
echo "bench N=200 x 1000 frames"; python3 "$R/bench.py" --features 200 --steps 1000 --warmup 10 --no-cpu-baseline > "$O/${P}_bench_n200_1000frames.json" 2>> "$O/bench.err"
echo "bench N=4000"; python3 "$R/bench.py" --features 4000 --steps 20 --warmup 3 --no-cpu-baseline --no-propagate-pass > "$O/${P}_bench_n4000.json" 2>> "$O/bench.err"
echo "rocprofv3 kernel stats (the timed run of the default bench line)"; rm -rf "$O/prof"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o ks -- python3 "$R/bench.py" --no-cpu-baseline --no-propagate-pass --no-live-traffic > "$O/${P}_bench_under_rocprof.json" 2>> "$O/bench.err"
cp "$O"/prof/ks_kernel_stats.csv "$O/${P}_kernel_stats.csv"
python3 "$R/tools/trace_timeline.py" "$O/prof/ks_kernel_trace.csv" > "$O/${P}_step_timeline.txt"
echo "rocprofv3 kernel stats WITH the streaming P-propagate pass (k_propagate_streaming row)"; rm -rf "$O/prof"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o ks -- python3 "$R/bench.py" --steps 40 --warmup 5 --no-cpu-baseline --no-live-traffic > "$O/${P}_bench_under_rocprof_propagate.json" 2>> "$O/bench.err"
cp "$O"/prof/ks_kernel_stats.csv "$O/${P}_kernel_stats_with_propagate_pass.csv"
echo "N=200 and N=4000 stats + timelines"; rm -rf "$O/prof"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o ks -- python3 "$R/bench.py" --features 200 --steps 300 --warmup 10 --no-cpu-baseline --no-propagate-pass --no-live-traffic > /dev/null 2>> "$O/bench.err"
cp "$O"/prof/ks_kernel_stats.csv "$O/${P}_kernel_stats_n200.csv"
python3 "$R/tools/trace_timeline.py" "$O/prof/ks_kernel_trace.csv" > "$O/${P}_step_timeline_n200.txt"
rm -rf "$O/prof"
echo "N=32 (the reference's operating point) bench line, stats + timeline"
python3 "$R/bench.py" --features 32 --steps 2000 --warmup 50 --no-cpu-baseline --no-live-traffic > "$O/${P}_bench_n32.json" 2>> "$O/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o ks -- python3 "$R/bench.py" --features 32 --steps 500 --warmup 20 --no-cpu-baseline --no-propagate-pass --no-live-traffic > /dev/null 2>> "$O/bench.err"
cp "$O"/prof/ks_kernel_stats.csv "$O/${P}_kernel_stats_n32.csv"
python3 "$R/tools/trace_timeline.py" "$O/prof/ks_kernel_trace.csv" > "$O/${P}_step_timeline_n32.txt"
rm -rf "$O/prof"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o ks -- python3 "$R/bench.py" --features 4000 --steps 10 --warmup 2 --no-cpu-baseline --no-propagate-pass --no-live-traffic > /dev/null 2>> "$O/bench.err"
cp "$O"/prof/ks_kernel_stats.csv "$O/${P}_kernel_stats_n4000.csv"
python3 "$R/tools/trace_timeline.py" "$O/prof/ks_kernel_trace.csv" > "$O/${P}_step_timeline_n4000.txt"
echo "PMC passes"; rm -rf "$O/pmc"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc/pmc_fetch" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic > /dev/null 2>> "$O/bench.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc/pmc_write" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic > /dev/null 2>> "$O/bench.err"
cd "$R" && python3 tools/pmc_summary.py gpurun_out/$P/pmc gpurun_out/$P/${P}_pmc_traffic.json
echo "MFMA utilisation passes (SQ / GRBM counters, counters + kernel trace only)"
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_mfma" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-propagate-pass --no-live-traffic > /dev/null 2>> "$O/bench.err"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$O/pmc_mfma2" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-propagate-pass --no-live-traffic > /dev/null 2>> "$O/bench.err"
cd "$R" && python3 tools/pmc_mfma.py gpurun_out/$P gpurun_out/$P/${P}_pmc_mfma.json > "$O/${P}_pmc_mfma.txt"
echo "configs[4] resize cadence on one GPU (N = 4000, an event every 10 frames so that a short run holds several)"
python3 "$R/bench.py" --features 4000 --steps 40 --warmup 3 --no-cpu-baseline --no-propagate-pass --no-live-traffic --resize-every 10 > "$O/${P}_bench_n4000_resize.json" 2>> "$O/bench.err"
rm -rf "$O/prof" "$O/pmc" "$O/pmc_mfma" "$O/pmc_mfma2"
fi
if [ "$STAGE" = B ]; then
echo "diagonal factor + column-chain microbenchmarks"
"$R/tools/diag_bench" > "$O/${P}_diag_bench.txt"
"$R/tools/chain_latency" > "$O/${P}_chain_latency.txt"
echo "per-kernel HIP-event times, the persistent chain kernel's task trace, the N = 200 accuracy decomposition"
python3 "$R/tools/kernel_ms.py" 1000 40 2>/dev/null > "$O/${P}_kernel_ms_n1000.txt"
python3 "$R/tools/chain_trace.py" 1000 2>/dev/null > "$O/${P}_chain_persistent_trace_final.txt"
python3 "$R/tools/n200_chunk_accuracy.py" 2>/dev/null | grep EKF_OPT > "$O/${P}_n200_chunks.txt"
python3 "$R/tools/n200_error_source.py" 2>/dev/null | tail -4 > "$O/${P}_n200_error_source.txt"
echo "sharded step on one rank over RCCL (world 1, forced collectives), plain row panel vs symmetric own block"
for sym in 0 1; do EKF_SHARD_SYM=$sym python3 "$R/tools/shard_world1.py" 1000 60 2>/dev/null | tail -1 > "$O/${P}_shard_world1_nccl_sym$sym.json"; done
python3 "$R/tools/shard_world1.py" 4000 12 2>/dev/null | tail -1 > "$O/${P}_shard_world1_nccl_n4000.json"
EKF_SHARD_DIST_CHAIN=0 python3 "$R/tools/shard_world1.py" 4000 12 2>/dev/null | tail -1 > "$O/${P}_shard_world1_nccl_n4000_replicated_chain.json"
EKF_SHARD_DIST_CHAIN=0 EKF_TD_MAX_BLOCKS=256 python3 "$R/tools/shard_world1.py" 4000 12 2>/dev/null | tail -1 > "$O/${P}_shard_world1_nccl_n4000_tdmax256.json"
echo "rccl_smoke rehearsal (gloo, 2 / 3 / 4 ranks on the one GPU)"
: > "$O/${P}_rccl_smoke_gloo.txt"
for g in 2 3 4; do python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $g --master-addr 127.0.0.1 --master-port 2954$g "$R/tools/rccl_smoke.py" --backend gloo 2>&1 | grep "rccl_smoke" >> "$O/${P}_rccl_smoke_gloo.txt"; done
echo "bench.py --gpus 5 rehearsal over gloo (five ranks on the one GPU: launcher + 5 ranks = the six processes the box allows)"
EKF_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port 29561 "$R/bench.py" --gpus 5 --steps 10 --warmup 2 2>/dev/null | grep "^{" > "$O/${P}_bench_gloo_5ranks.json" || true
echo "knob A/B at N = 1000 (exact fp32 against the default, chunk plans) and accuracy of the same knobs"
python3 "$R/tools/knob_ab.py" "" "EKF_CHAIN_FUSED_DIAG=0" "EKF_CHAIN_DEFER=0" "EKF_SU_TAIL=0" "EKF_SYRK_STAGGER=0,0" "EKF_FUSE_SPLIT=0" "EKF_CHAIN_FUSED_DIAG=0 EKF_CHAIN_DEFER=0 EKF_SU_TAIL=0 EKF_SYRK_STAGGER=0,0 EKF_FUSE_SPLIT=0" "EKF_CHAIN_PERSISTENT=1" "EKF_SPLIT_BF16=0" "EKF_W_RECOMPUTE=0" "EKF_CHUNKS=3,7,12,16" "EKF_CHUNKS=2,6,11,16" "EKF_RESERVED_CUS=24" "EKF_RESERVED_CUS=48" 2>/dev/null > "$O/${P}_knob_ab_n1000.txt"
KNOB_N=4000 KNOB_FRAMES=40 python3 "$R/tools/knob_ab.py" "" "EKF_CHAIN_FUSED_DIAG=0" "EKF_CHAIN_FUSED_DIAG=0 EKF_CHAIN_DEFER=0 EKF_SU_TAIL=0 EKF_SYRK_STAGGER=0,0" "EKF_SPLIT_BF16=0" 2>/dev/null > "$O/${P}_knob_ab_n4000.txt"
KNOB_N=2000 KNOB_FRAMES=80 python3 "$R/tools/knob_ab.py" "" "EKF_CHAIN_FUSED_DIAG=0" "EKF_CHAIN_FUSED_DIAG=0 EKF_CHAIN_DEFER=0 EKF_SU_TAIL=0 EKF_SYRK_STAGGER=0,0" "EKF_SPLIT_BF16=0" 2>/dev/null > "$O/${P}_knob_ab_n2000.txt"
python3 "$R/tools/acc_knobs.py" "" "EKF_SPLIT_BF16=0" "EKF_W_RECOMPUTE=0" "EKF_SPLIT_BF16=0 EKF_CHUNKS=3,7,16" 2>/dev/null > "$O/${P}_accuracy_knobs_n1000.txt"
fi
if [ "$STAGE" = C ]; then
echo "N = 4000 against the fp64 oracle, every entry (two frames)"
python3 "$R/tools/n4000_oracle_parity.py" --hip 2>/dev/null | tail -2 > "$O/${P}_n4000_oracle_parity.txt"
fi
if [ "$STAGE" = D ]; then
echo "long all-measured runs on the default path (positivity of the fp32 covariance)"
for nf in "1000 3000" "2000 2500" "4000 1200"; do set -- $nf; python3 "$R/tools/long_run.py" $1 $2 2>/dev/null | tail -2 > "$O/${P}_long_run_n$1.txt" || true; done
fi
if [ "$STAGE" = E ]; then
echo "CPU baseline with the measured single-thread dense frame (--cpu-full)"
python3 "$R/bench.py" --steps 20 --warmup 5 --no-propagate-pass --no-live-traffic --cpu-full > "$O/${P}_bench_cpu_full.json" 2>> "$O/bench.err"
fi
ls "$O"
