#!/bin/bash
# round 6: the soft forward under the round-4 library (compact two-pass strip lists) and the current one (one-pass
# two-ended regions) on one box: kernel times, then L2 hit / miss and HBM bytes of k_mesh_soft_forward per launch
export TMPDIR=/tmp
CASES="soft" REPS=2 bash tools/ab_libs.sh tools/lib_r04_final.so mvtn_amd/libmvr_hip.so
for lib in tools/lib_r04_final.so mvtn_amd/libmvr_hip.so; do
  for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
    d=gpurun_out/r06_softpmc_$(basename $lib .so)_$(echo $c | tr ' ' '_')
    MVR_LIB=$lib timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --workload soft_fwdbwd --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
    python3 - "$d" "$lib" "$c" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mesh_soft_forward" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = {}
for k, v in acc.items():
    # one row per (dispatch, counter[, dimension]): sum over rows / launches
    n[k] = sum(v)
print(sys.argv[2], {k: "%.4g" % x for k, x in n.items()}, "(summed over the profiled launches)")
PY
  done
done
