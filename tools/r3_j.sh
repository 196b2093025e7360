cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for i in 1 2 3; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('driver-like: step %.1f us value %.3f M kernel %.1f us two_streams %.1f' % (d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['kernel_ms']*1e3, d['two_streams']['ms_per_step']*1e3))"
done
python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('default 1000 steps: step %.1f us value %.3f M kernel %.1f us two_streams %.1f' % (d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['kernel_ms']*1e3, d['two_streams']['ms_per_step']*1e3))"
python3 bench.py --workload mesh_fwdbwd --no-cpu-baseline --steps 300 --warmup 50 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('C3 eager %.1f us graph %.1f us raster %.1f bwd %.1f' % (d['ms_per_step']*1e3, d['hip_graph_replay']['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline']['backward_kernel_ms']*1e3))"
python3 bench.py --workload c5 --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('C5 64 objects on one GPU: step %.1f us value %.3f M kernel %.1f' % (d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['kernel_ms']*1e3)); print(d['config']['workload'])"
