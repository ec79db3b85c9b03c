#!/bin/bash
# Step time against the number of CUs kept free for the factorisation chain (EKF_RESERVED_CUS tuning knob).
for r in 16 24 28 30 32 34 36 40 48; do
  echo -n "reserved $r: "
  EKF_RESERVED_CUS=$r python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
