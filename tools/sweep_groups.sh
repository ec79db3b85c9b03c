#!/bin/bash
# throughput vs number of pipeline column groups (EKF_OPT_PIPELINE = k)
for g in 0 2 3 4 5 6; do
  python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-propagate-pass --pipeline $g 2>/dev/null > /tmp/sweep_$g.json
  python - "$g" <<'PY'
import json, sys
d = json.load(open(f"/tmp/sweep_{sys.argv[1]}.json"))
print("groups", sys.argv[1], d["value"], d["ms_per_step"], d["run_sane"])
PY
done
