cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3g
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -12
timeout 600 python3 bench.py --workload soft_fwdbwd --steps 50 --warmup 5 --no-cpu-baseline 2>gpurun_out/r3g/err.txt | tail -1 > gpurun_out/r3g/soft.json
python3 -c "
import json
d=json.load(open('gpurun_out/r3g/soft.json'))
print('soft step %.1f us value %.4f M fwd kernel %.1f us bwd kernel %.1f us' % (d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['kernel_ms']*1e3, d['roofline'].get('backward_kernel_ms',0)*1e3))
"
tail -3 gpurun_out/r3g/err.txt
