#!/bin/bash
for s in 6 8 9 10 11 12 13; do
  EKF_SPLIT16=$s python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-propagate-pass 2>/dev/null > /tmp/sw_$s.json
  python - "$s" <<'PY'
import json, sys
d = json.load(open(f"/tmp/sw_{sys.argv[1]}.json"))
print("first group", sys.argv[1], "/16:", d["value"], d["ms_per_step"], d["run_sane"])
PY
done
