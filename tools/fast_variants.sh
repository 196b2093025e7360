# developer scratch: kernel time of k_mesh_raster_fast for a few -D variants (rocprofv3 kernel stats, C2)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "$@"; do
  d=""; [ "$v" != base ] && d="$v"
  python3 mvtn_amd/build.py /tmp/libvar.so $d > /dev/null || { echo "build failed: $v"; continue; }
  export MVR_LIB=/tmp/libvar.so
  rm -rf /tmp/var_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/var_prof -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-pipelined > /tmp/var_bench.json 2>/dev/null
  echo "== $v  $(MVR_DEBUG_OCC=1 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined 2>&1 | grep '\[mvr\]' | head -1 | cut -c1-75)"
  find /tmp/var_prof -name "*kernel_stats.csv" | head -1 | xargs grep "raster\|prepass" | cut -d, -f1-4 | sed 's/(.*)//'
done
