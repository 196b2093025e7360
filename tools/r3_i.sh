cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in tools/var_b2.so tools/var_b3.so tools/var_b4.so tools/var_b5.so; do
MVR_LIB=$lib timeout 600 python3 bench.py --workload soft_fwdbwd --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$lib soft step %.1f us fwd kernel %.1f us bwd kernel %.1f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline'].get('backward_kernel_ms',0)*1e3))
"
done
