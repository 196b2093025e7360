cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3f
timeout 600 python3 bench.py --workload soft_fwdbwd --steps 50 --warmup 5 --cpu-views 4 2>gpurun_out/r3f/err.txt | tail -1 > gpurun_out/r3f/soft.json
python3 -c "
import json
d=json.load(open('gpurun_out/r3f/soft.json'))
print('soft step %.1f us value %.4f M fwd kernel %.1f us bwd kernel %.1f us parity %s graph %s' % (d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['kernel_ms']*1e3, d['roofline'].get('backward_kernel_ms',0)*1e3, d['max_pixel_err_vs_oracle'], d.get('hip_graph_replay')))
"
tail -5 gpurun_out/r3f/err.txt
timeout 600 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5
