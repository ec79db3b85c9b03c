#!/bin/bash
# HBM bytes per launch of the downdate kernel (FETCH_SIZE x 2 + WRITE_SIZE, separate PMC passes): tools/pmc_downdate.sh <tag>
R="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/pmc_dd"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_dd/pmc_fetch" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-propagate-pass > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_dd/pmc_write" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-propagate-pass > /dev/null 2>&1
cd "$R" && python3 tools/pmc_summary.py gpurun_out/pmc_dd gpurun_out/pmc_dd_$1.json | grep "k_gemm_mfma<2"
