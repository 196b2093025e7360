# developer scratch: soft forward / backward kernel times of tools/libhead.so and tools/var_*.so (C3 shape), interleaved
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do for lib in tools/libhead.so tools/var_*.so; do
  MVR_LIB=$lib timeout 200 python3 bench.py --workload soft_fwdbwd --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib step %.3f ms fwd %.3f bwd %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('backward_kernel_ms',0)))"
done; done
