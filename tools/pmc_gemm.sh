#!/bin/bash
# PMC passes over tools/gemm_bench (per-dispatch counters of the tile GEMM variants)
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d "$R/gpurun_out/pmc_gemm_$i" -o p --output-format csv -- "$R/tools/gemm_bench" > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.OrderedDict()
for f in sorted(glob.glob(R + "/gpurun_out/pmc_gemm_*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"][:60])
        agg.setdefault(k, {})[r["Counter_Name"]] = agg.setdefault(k, {}).get(r["Counter_Name"], 0) + float(r["Counter_Value"])
for (d, name), c in list(agg.items())[:40]:
    print(d, name, " ".join(f"{k}={v:.3g}" for k, v in c.items()))
PY
