#!/bin/bash
# kernel-trace timeline of one EKF step: tools/timeline.sh <tag> [bench.py arguments...]
R="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
tag=${1:-default}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/tl"
rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/tl" -o kt -- python3 "$R/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-propagate-pass "$@" > /dev/null 2>&1
f=$(ls "$R"/gpurun_out/tl/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -z "$f" ] && f=$(ls "$R"/gpurun_out/tl/*kernel_trace.csv | head -1)
python3 "$R/tools/trace_timeline.py" $f > "$R/gpurun_out/timeline_${tag}.txt"
rm -rf "$R/gpurun_out/tl"
