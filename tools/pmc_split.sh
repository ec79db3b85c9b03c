#!/bin/bash
# PMC counters of the split-bf16 downdate kernel inside the bench run
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d "$R/gpurun_out/pmc_split_$i" -o p --output-format csv -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-propagate-pass --split-bf16 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(R + "/gpurun_out/pmc_split_*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "syrk_bf16x3" in r["Kernel_Name"] or "k_gemm_mfma<2" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    g = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8 if c.get("GRBM_GUI_ACTIVE") else 1
    for name, v in c.items():
        m = sum(v) / len(v)
        print(f"   {name:32s} mean {m:14.4g}   per-xcd-cycle {m / g / 8:10.3f}")
PY
