#!/bin/bash
# One step of bench.py as a kernel timeline: tools/step_timeline.sh <name under gpurun_out/> <bench.py arguments...>
# (environment settings are passed through; run on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p "$O"; rm -rf "$O/prof"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o ks -- python3 "$R/bench.py" --no-cpu-baseline --no-propagate-pass --no-live-traffic "$@" > "$O/bench.json" 2> "$O/bench.err"
python3 "$R/tools/trace_timeline.py" "$O/prof/ks_kernel_trace.csv" > "$O/timeline.txt"
cp "$O/prof/ks_kernel_stats.csv" "$O/kernel_stats.csv" 2>/dev/null || true
rm -rf "$O/prof"
cat "$O/timeline.txt"
python3 -c "
import json,sys; d=json.load(open('$O/bench.json')); print('value', d['value'], 'ms_per_step', d['ms_per_step'])"
