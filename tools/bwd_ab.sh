# developer scratch: C3 step / backward kernel of tools/libhead.so and tools/var_*.so, interleaved
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2 3; do for lib in tools/libhead.so tools/var_*.so; do
  MVR_LIB=$lib timeout 100 python3 bench.py --workload mesh_fwdbwd --no-cpu-baseline --no-pipelined --steps 500 --warmup 100 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib fwdbwd eager %.1f raster %.1f bwd %.1f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline']['backward_kernel_ms']*1e3))"
done; done
