# developer scratch: k_mesh_prepass time for -D variants (C2 forward under rocprofv3)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "$@"; do
  d=""; [ "$v" != base ] && d="$v"
  python3 mvtn_amd/build.py /tmp/libvar.so $d > /dev/null || { echo "build failed: $v"; continue; }
  export MVR_LIB=/tmp/libvar.so
  rm -rf /tmp/var_prof
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/var_prof -- python3 bench.py --workload ${PP_WORKLOAD:-mesh_fwd} --steps 40 --warmup 10 --no-cpu-baseline --no-pipelined > /tmp/var_bench.json 2>/dev/null
  echo "== $v"
  python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("/tmp/var_prof/**/*kernel_stats.csv",recursive=True)[0])):
    if "k_mesh" in r["Name"]: print("   %-40s %4s  avg %.1f us  min %.1f  max %.1f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
