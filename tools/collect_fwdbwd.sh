# developer scratch: the C3 training-step bench lines of one lease — C++ operator / Python node, warm (default) / cold (20 steps,
# no pre-warm) — into gpurun_out/$1/bench_mesh_fwdbwd_$2*.json.   bash tools/collect_fwdbwd.sh r04_c leaseA
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/$1; tag=$2; mkdir -p $out
run() { timeout 300 python3 bench.py --workload mesh_fwdbwd --no-cpu-baseline "$@" 2>/dev/null | tail -1; }
run > $out/bench_mesh_fwdbwd_$tag.json
MVR_TORCH_OP=0 run > $out/bench_mesh_fwdbwd_${tag}_python_node.json
run --steps 20 --warmup 5 --prewarm 0 > $out/bench_mesh_fwdbwd_${tag}_cold.json
MVR_TORCH_OP=0 run --steps 20 --warmup 5 --prewarm 0 > $out/bench_mesh_fwdbwd_${tag}_cold_python_node.json
for f in $out/bench_mesh_fwdbwd_$tag*.json; do python3 -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], 'value %.0f  %.1f us/step  host %.1f us  graph %.1f  1-thread %.1f' % (d['value'], d['ms_per_step']*1e3, d['host_enqueue_ms_per_step']*1e3, d.get('hip_graph_replay',{}).get('ms_per_step',0)*1e3, d.get('eager_autograd_single_thread',{}).get('ms_per_step',0)*1e3))"; done
