"""Developer tool: per-step timeline from a rocprofv3 --kernel-trace CSV: kernel order, durations and the idle gap
before each kernel (median over the steps).  Usage: python tools/trace_gaps.py <kernel_trace.csv> <first kernel substr>"""
import csv
import sys
from collections import defaultdict

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "k_look_at"
steps, cur = [], []
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    if first in name and cur and first in cur[0][0] and not any(first in c[0] for c in cur[1:]) and len(cur) > 1:
        steps.append(cur)
        cur = []
    cur.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
steps = steps[len(steps) // 2:]          # the timed half
sig = defaultdict(list)
for st in steps:
    key = tuple(k[0] for k in st)
    sig[key].append(st)
key, group = max(sig.items(), key=lambda kv: len(kv[1]))
print("steps with the dominant kernel sequence: %d of %d" % (len(group), len(steps)))
tot = np.median([st[-1][2] - st[0][1] for st in group]) / 1e3
print("median first-kernel-start -> last-kernel-end: %.1f us" % tot)
period = np.median(np.diff([st[0][1] for st in group])) / 1e3
print("median step period: %.1f us" % period)
for i, name in enumerate(key):
    dur = np.median([st[i][2] - st[i][1] for st in group]) / 1e3
    gap = np.median([st[i][1] - st[i - 1][2] for st in group]) / 1e3 if i else 0.0
    print("  %-50s gap %7.1f us   dur %7.1f us" % (name[:50], gap, dur))
