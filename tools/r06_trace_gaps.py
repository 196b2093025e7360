"""Developer tool (round 6): per-step period and inter-kernel gaps of the point path from a rocprofv3 --kernel-trace CSV.
    python3 tools/r06_trace_gaps.py <dir>"""
import csv, glob, sys
import numpy as np
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
names = [r[2] for r in rows]
idx = [i for i, n in enumerate(names) if n == "k_points_raster_shade"]
print("kernels", len(rows), "point raster launches", len(idx))
for lo, hi, label in ((len(idx) // 2 - 400, len(idx) // 2 - 100, "middle"), (len(idx) - 350, len(idx) - 50, "late")):
    sel = idx[lo:hi]
    st = np.array([rows[i][0] for i in sel], float); en = np.array([rows[i][1] for i in sel], float)
    period = np.diff(st) / 1e3
    dur = (en - st) / 1e3
    # the kernel before each raster launch (the pre-pass) and the gap between them
    pre_gap = np.array([rows[i][0] - rows[i - 1][1] for i in sel], float) / 1e3
    pre_dur = np.array([rows[i - 1][1] - rows[i - 1][0] for i in sel], float) / 1e3
    gap_after = np.array([rows[i + 1][0] - rows[i][1] for i in sel[:-1]], float) / 1e3
    print("%s: period %.1f us (median %.1f), raster %.1f, kernel before it %s %.1f, gap pre->raster %.2f, gap raster->next %.2f (negative = overlap)" % (
        label, period.mean(), np.median(period), dur.mean(), rows[sel[0] - 1][2], pre_dur.mean(), pre_gap.mean(), gap_after.mean()))
