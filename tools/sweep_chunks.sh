#!/bin/bash
# throughput vs chunk pattern of the factorisation (EKF_CHUNKS = chunk ends in block steps, N = 1000: 16 steps)
run() {
  python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-propagate-pass 2>/dev/null > /tmp/sweep_c.json
  python - "$1" <<'PY'
import json, sys
d = json.load(open("/tmp/sweep_c.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], d["run_sane"])
PY
}
for c in ${CHUNK_LIST:-"4,10,16" "4,10,14,16" "4,9,13,16" "3,8,13,16" "4,8,12,16" "4,10,13,16" "5,10,14,16" "4,9,14,16" "3,7,11,14,16" "4,8,12,14,16" "2,6,10,14,16"}; do
  export EKF_CHUNKS=$c
  run "chunks $c"
done
