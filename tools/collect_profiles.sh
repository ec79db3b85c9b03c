set -e
R="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" > "$R/gpurun_out/r1_bench.json" 2> "$R/gpurun_out/r1_bench.err"
python3 "$R/bench.py" --features 200 --steps 1000 --warmup 10 > "$R/gpurun_out/r1_bench_n200.json" 2>> "$R/gpurun_out/r1_bench.err"
rm -rf "$R/gpurun_out/prof_r1b"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_r1b" -o ks -- python3 "$R/bench.py" --no-cpu-baseline --no-propagate-pass > "$R/gpurun_out/r1_bench_under_rocprof.json" 2>> "$R/gpurun_out/r1_bench.err"
rm -rf "$R/gpurun_out/pmc_r1b"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_r1b/pmc_fetch" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>> "$R/gpurun_out/r1_bench.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_r1b/pmc_write" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>> "$R/gpurun_out/r1_bench.err"
cd "$R" && python3 tools/pmc_summary.py gpurun_out/pmc_r1b gpurun_out/r1_pmc_traffic.json
ls gpurun_out/prof_r1b
