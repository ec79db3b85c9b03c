"""k_propagate_streaming under rocprofv3 --kernel-trace: duration of every launch against the idle time of the GPU in front
of it (the kernel averages ~104 us in a profiled run and 49 us -- 0.73 of the HBM peak -- in an
un-profiled one).  usage: python3 tools/propagate_trace.py <ks_kernel_trace.csv>

The tracer makes every dispatch wait for the completion signal of the one before it, so inside a profiled run the GPU idles
between kernels; this table shows what that does to a bandwidth-bound kernel that follows a stretch of tiny launches."""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ekf::", "").replace("ekf::", "")))
rows.sort()
out = []
busy_end = 0.0
win = []            # (start, end) of the kernels in the last 2 ms
for k, (s, e, name) in enumerate(rows):
    if name.startswith("k_propagate_streaming"):
        gap = (s - busy_end) / 1e3 if busy_end else float("nan")
        lo = s - 2e6
        busy = sum(min(b, s) - max(a, lo) for a, b in win if b > lo) / 2e6
        out.append((s, (e - s) / 1e3, gap, busy))
    busy_end = max(busy_end, e)
    win.append((s, e))
    win = [w for w in win if w[1] > s - 2e6]
print("# launch  duration_us  idle_in_front_us  gpu_busy_share_of_the_2_ms_in_front")
for i, (s, d, gap, busy) in enumerate(out):
    print(f"{i:5d} {d:10.1f} {gap:12.1f} {busy:10.3f}")
import statistics as st
d = [o[1] for o in out]
if d:
    print(f"# {len(d)} launches: mean {st.mean(d):.1f} us, median {st.median(d):.1f}, min {min(d):.1f}, max {max(d):.1f}")
    fast = [o for o in out if o[1] < 60]
    slow = [o for o in out if o[1] >= 60]
    for tag, grp in (("< 60 us", fast), (">= 60 us", slow)):
        if grp:
            print(f"# launches {tag}: {len(grp)}, mean busy share in front {st.mean(g[3] for g in grp):.3f}, mean idle in front {st.mean(g[2] for g in grp):.1f} us")
# which kernels run WHILE the slow launches run (overlap > 2 us)
print("# kernels overlapping the first slow launches:")
shown = 0
for (s, d, gap, busy) in out:
    if d < 60 or shown >= 6:
        continue
    e = s + d * 1e3
    ov = [(min(e, b) - max(s, a), nm) for (a, b, nm) in rows if b > s and a < e and not nm.startswith("k_propagate_streaming")]
    ov = [(o / 1e3, nm) for o, nm in ov if o > 2e3]
    print(f"#   launch at {s / 1e3:.1f} us, {d:.1f} us: " + (", ".join(f"{nm} ({o:.0f} us)" for o, nm in ov) or "none"))
    shown += 1
