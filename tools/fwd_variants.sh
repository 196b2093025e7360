# developer scratch: step / raster-kernel time of the mesh forward for -D variants (C2 and C5), and fast-vs-exact parity
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "$@"; do
  d=""; [ "$v" != base ] && d="$v"
  python3 mvtn_amd/build.py /tmp/libvar.so $d > /dev/null || { echo "build failed: $v"; continue; }
  export MVR_LIB=/tmp/libvar.so
  echo "== $v"
  for w in mesh_fwd c5; do
    timeout 100 python3 bench.py --workload $w --no-cpu-baseline --no-pipelined --steps 500 --warmup 100 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   $w step %.1f kernel %.1f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
  done
  timeout 200 python3 tools/fast_check.py 2>&1 | tail -3
done
