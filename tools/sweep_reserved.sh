#!/bin/bash
for r in 8 16 24 32 48; do for g in 2 4; do
  EKF_RESERVED_CUS=$r python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-propagate-pass --pipeline $g 2>/dev/null > /tmp/sr.json
  python - "$r" "$g" <<'PY'
import json, sys
d = json.load(open("/tmp/sr.json"))
print("reserved", sys.argv[1], "groups", sys.argv[2], ":", d["value"], d["ms_per_step"], d["run_sane"])
PY
done; done
