#!/usr/bin/env python3
"""tests/golden/parity_bounds.json from a parity log of the GPU tests.

  EKF_PARITY_LOG=gpurun_out/parity_log.jsonl python -m pytest tests -m gpu      (on the MI355X box)
  python tools/update_parity_bounds.py gpurun_out/parity_log.jsonl              (here)

Per call site of tests/helpers.bound() (test node id + label) the MAXIMUM error measured in that run is recorded;
bound() then holds the site to 10x that value (see its docstring).  The log itself is kept under profiles/ so the
numbers can be audited."""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    log = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_log.jsonl")
    sites = collections.OrderedDict()
    for line in open(log):
        r = json.loads(line)
        sites[r["site"]] = max(sites.get(r["site"], 0.0), r["value"])
    out = {"generated_by": "tools/update_parity_bounds.py", "source_log": os.path.relpath(log, ROOT),
           "rule": "bound = min(ceiling in the test, max(10 x measured, floor)), floor 3e-7 (fp32 sites) / 1e-14 (fp64 sites)",
           "sites": sites}
    path = os.path.join(ROOT, "tests", "golden", "parity_bounds.json")
    json.dump(out, open(path, "w"), indent=1)
    print(f"{len(sites)} call sites -> {path}")


if __name__ == "__main__":
    main()
