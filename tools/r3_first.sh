# round 3, first GPU call: the whole GPU suite, then the forward issued in 1 / 2 / 3 / 4 pieces at C2 and tier L
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3a/pytest.txt
cat gpurun_out/r3a/pytest.txt
for p in 0 1 2 3 4; do
  for w in mesh_fwd c5_per_gpu; do
    timeout 200 python3 bench.py --workload $w --forward-pieces $p --no-cpu-baseline --steps 1000 --warmup 100 2>>gpurun_out/r3a/err.txt | tail -1 > gpurun_out/r3a/bench_${w}_p$p.json
    python3 - <<PY
import json
d=json.load(open("gpurun_out/r3a/bench_${w}_p$p.json"))
print("$w pieces=$p step %.1f us  value %.3f M  kernel %.1f us  one_piece %.1f us  two_streams %.1f us" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["roofline"]["kernel_ms"]*1e3, d.get("one_piece",{}).get("ms_per_step",0)*1e3, d.get("two_streams",{}).get("ms_per_step",0)*1e3))
PY
  done
done
timeout 300 python3 bench.py --workload mesh_fwdbwd --no-cpu-baseline --steps 500 --warmup 100 2>>gpurun_out/r3a/err.txt | tail -1 > gpurun_out/r3a/bench_fwdbwd.json
python3 -c "
import json
d=json.load(open('gpurun_out/r3a/bench_fwdbwd.json'))
print('fwdbwd eager step %.1f us, graph %s, raster %.1f bwd %.1f' % (d['ms_per_step']*1e3, d.get('hip_graph_replay'), d['roofline']['kernel_ms']*1e3, d['roofline'].get('backward_kernel_ms',0)*1e3))
"
timeout 120 python3 bench.py --steps 20 --warmup 5 2>>gpurun_out/r3a/err.txt | tail -1 > gpurun_out/r3a/bench_driver_like.json
python3 -c "
import json
d=json.load(open('gpurun_out/r3a/bench_driver_like.json'))
print('driver-like: step %.1f us value %.3f M kernel %.1f us err %s cpu %s' % (d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['kernel_ms']*1e3, d['max_pixel_err_vs_oracle'], d['cpu_baseline']['value']))
"
tail -5 gpurun_out/r3a/err.txt
