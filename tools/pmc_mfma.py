"""MFMA utilisation of every kernel of the bench step from rocprofv3 SQ / GRBM counters (VERDICT r4 #3a).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
              -d $R/gpurun_out/r5/pmc_mfma -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-propagate-pass --no-live-traffic
    rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
              --kernel-trace --output-format csv -d $R/gpurun_out/r5/pmc_mfma2 -- python3 $R/bench.py ... (same)
    python3 tools/pmc_mfma.py gpurun_out/r5 profiles/r5_pmc_mfma.json

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x the kernel's GPU cycles); the kernel's GPU cycles are
GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the counter over the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back) -- i.e. the share of
the chip's matrix-pipe cycles that had an MFMA in flight while the kernel ran (a kernel on a CU-masked stream, or one that
shares the chip with another stream's kernel, is counted against the WHOLE chip, and the counters of concurrent kernels
overlap: the passes serialise kernels, so each row is the kernel alone on the chip).  effective_clock_ghz = GPU cycles /
the kernel's duration from the kernel trace of the same pass."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha():
    h = hashlib.sha1()
    d = os.path.join(ROOT, "ekf-monoslam_for_3d-reconstruction_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hpp", ".hip")):
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def short(k):
    return re.sub(r"\(.*", "", k).replace("void ekf::", "").replace("ekf::", "")


out = {}
for sub in ("pmc_mfma", "pmc_mfma2"):
    files = sorted(glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True))
    if not files:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[-1])):
        per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(f"{src}/{sub}/**/*kernel_trace.csv", recursive=True))[-1:]:
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, cs in per.items():
        if not k.startswith("k_"):
            continue
        o = out.setdefault(k, {})
        for cn, v in cs.items():
            o[cn + "_mean"] = sum(v) / len(v)
            o.setdefault("dispatches", len(v))
        if k in dur and sub == "pmc_mfma":
            o["duration_us_mean_in_this_pass"] = sum(dur[k]) / len(dur[k]) / 1e3
for k, o in out.items():
    gui = o.get("GRBM_GUI_ACTIVE_mean")
    busy = o.get("SQ_VALU_MFMA_BUSY_CYCLES_mean")
    if gui and busy is not None:
        cyc = gui / 8.0
        o["gpu_cycles"] = cyc
        o["mfma_busy"] = busy / (4.0 * 256.0 * cyc)
        if o.get("duration_us_mean_in_this_pass"):
            o["effective_clock_ghz"] = cyc / (o["duration_us_mean_in_this_pass"] * 1e3)
json.dump({"note": __doc__, "csrc_sha16": csrc_sha(), "kernels": out}, open(dst, "w"), indent=1)
for k, o in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES_mean", 0))[:12]:
    print(f"{k:44s} mfma_busy {o.get('mfma_busy', float('nan')):6.3f}  clock {o.get('effective_clock_ghz', float('nan')):5.2f} GHz  "
          f"{o.get('duration_us_mean_in_this_pass', float('nan')):8.1f} us")
