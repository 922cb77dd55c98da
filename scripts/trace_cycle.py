"""Summarise a rocprofv3 kernel trace CSV: per-kernel totals and the timeline (kernel, start offset, duration,
gap to the previous kernel) of the LAST full V-cycle window found between two residual_norm kernels."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "residual_norm_partials" in n or "residual_norm_slices" in n]
# windows between consecutive norm kernels that hold a whole V-cycle (many kernels); take a middle one
wins = [(idx[i], idx[i + 1]) for i in range(len(idx) - 1) if idx[i + 1] - idx[i] > 20]
a, b = wins[len(wins) // 2]
t0 = int(rows[a]["End_Timestamp"])
prev_end = t0
busy = 0
print(f"window: {b - a} kernels, {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us wall")
for r in rows[a + 1 : b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    short = r["Kernel_Name"].split("(")[0].replace("void gmgk::", "")[:40]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8}  {short}")
    busy += e - s
    prev_end = e
print(f"busy {busy / 1e3:.1f} us")
