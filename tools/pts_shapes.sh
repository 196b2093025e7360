# developer scratch: the point rasteriser of tools/libhead.so vs tools/var_*.so at a few shapes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for shape in "" "--batch 32" "--batch 8" "--image-size 128" "--image-size 448 --batch 4" "--views 12"; do
for rep in 1 2; do for lib in tools/libhead.so tools/var_*.so; do
  MVR_LIB=$lib timeout 100 python3 bench.py --workload points_fwd $shape --no-cpu-baseline --steps 400 --warmup 100 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$shape] $lib step %.1f kernel %.1f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
done; done; done
