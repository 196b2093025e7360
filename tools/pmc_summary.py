"""Developer tool: average the per-dispatch counter values of a rocprofv3 --pmc run per kernel.
    python tools/pmc_summary.py <dir with *_counter_collection.csv> [kernel-substring]"""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if sub in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in acc.items()}, indent=1))
