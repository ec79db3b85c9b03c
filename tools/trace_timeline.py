"""Print the per-kernel timeline of the last full EKF step found in a rocprofv3 kernel trace csv."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ekf::", "").replace("ekf::", "")
    return n[:40]
# find the last k_predict_camera / k_predict_fused -> next one window
idx = [i for i, r in enumerate(rows) if "k_predict_camera" in r["Kernel_Name"] or "k_predict_fused" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
print("step window %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
for r in rows[a:b]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    e = (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e-s:8.1f}  q{r.get('Queue_Id','?'):>3} {short(r['Kernel_Name'])}")
