#!/bin/bash
# throughput vs chunk pattern of the factorisation (EKF_CHUNKS = chunk ends in block steps, N = 1000: 16 steps),
# chain-stream CU masking and the number of reserved CUs
run() {
  python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-propagate-pass 2>/dev/null > /tmp/sweep_c.json
  python - "$1" <<'PY'
import json, sys
d = json.load(open("/tmp/sweep_c.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], d["run_sane"])
PY
}
for mask in 1 0; do
  for res in 16 32 48; do
    for c in "8,16" "4,10,16" "4,8,12,16" "2,8,16" "3,8,13,16" "2,6,11,16"; do
      export EKF_CHAIN_MASK=$mask EKF_RESERVED_CUS=$res EKF_CHUNKS=$c
      run "mask $mask reserved $res chunks $c"
    done
  done
done
