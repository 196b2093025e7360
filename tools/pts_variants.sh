# developer scratch: kernel times of the point path for -D variants (C4)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "$@"; do
  d=""; [ "$v" != base ] && d="$v"
  python3 mvtn_amd/build.py /tmp/libvar.so $d > /dev/null || { echo "build failed: $v"; continue; }
  export MVR_LIB=/tmp/libvar.so
  rm -rf /tmp/var_prof
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/var_prof -- python3 bench.py --workload points_fwd --steps 40 --warmup 10 --no-cpu-baseline > /tmp/var_bench.json 2>/dev/null
  echo "== $v"
  python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("/tmp/var_prof/**/*kernel_stats.csv",recursive=True)[0])):
    if "k_points" in r["Name"]: print("   %-40s %4s  %.1f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
