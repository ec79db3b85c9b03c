#!/bin/bash
# Round-2 evidence in one call on the GPU box: tools/collect_profiles.sh   (outputs under gpurun_out/r2/, copied to profiles/)
set -e
R="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
O="$R/gpurun_out/r2"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
echo "bench N=1000"; python3 "$R/bench.py" > "$O/r2_bench.json" 2> "$O/bench.err"
echo "bench N=200 x 1000 frames"; python3 "$R/bench.py" --features 200 --steps 1000 --warmup 10 --no-cpu-baseline > "$O/r2_bench_n200_1000frames.json" 2>> "$O/bench.err"
echo "bench N=4000"; python3 "$R/bench.py" --features 4000 --steps 20 --warmup 3 --no-cpu-baseline --no-propagate-pass > "$O/r2_bench_n4000.json" 2>> "$O/bench.err"
echo "rocprofv3 kernel stats"; rm -rf "$O/prof"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o ks -- python3 "$R/bench.py" --no-cpu-baseline --no-propagate-pass --no-live-traffic > "$O/r2_bench_under_rocprof.json" 2>> "$O/bench.err"
cp "$O"/prof/ks_kernel_stats.csv "$O/r2_kernel_stats.csv"
python3 "$R/tools/trace_timeline.py" "$O/prof/ks_kernel_trace.csv" > "$O/r2_step_timeline.txt"
echo "PMC passes"; rm -rf "$O/pmc"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc/pmc_fetch" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic > /dev/null 2>> "$O/bench.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc/pmc_write" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic > /dev/null 2>> "$O/bench.err"
cd "$R" && python3 tools/pmc_summary.py gpurun_out/r2/pmc gpurun_out/r2/r2_pmc_traffic.json
rm -rf "$O/prof" "$O/pmc"
ls "$O"
