cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3b
timeout 300 python3 -m pytest tests/test_gpu_round3.py -x -q --tb=short 2>&1 | tail -30
export MVR_PIECES_MODE=anyorder
for p in 1 2 4; do
  for w in mesh_fwd c5_per_gpu; do
    timeout 200 python3 bench.py --workload $w --forward-pieces $p --no-cpu-baseline --no-pipelined --steps 1000 --warmup 100 2>>gpurun_out/r3b/err.txt | tail -1 > gpurun_out/r3b/bench_${w}_p$p.json
    python3 - <<PY
import json
d=json.load(open("gpurun_out/r3b/bench_${w}_p$p.json"))
print("anyorder $w pieces=$p step %.1f us  value %.3f M  kernel %.1f us" % (d["ms_per_step"]*1e3, d["value"]/1e6, d["roofline"]["kernel_ms"]*1e3))
PY
  done
done
timeout 200 python3 bench.py --workload c5 --forward-pieces 1 --no-cpu-baseline --no-pipelined --steps 200 --warmup 20 2>>gpurun_out/r3b/err.txt | tail -1 > gpurun_out/r3b/c5_64_p1.json
timeout 200 python3 bench.py --workload c5 --forward-pieces 8 --no-cpu-baseline --no-pipelined --steps 200 --warmup 20 2>>gpurun_out/r3b/err.txt | tail -1 > gpurun_out/r3b/c5_64_p8.json
python3 -c "
import json
for p in (1,8):
    d=json.load(open('gpurun_out/r3b/c5_64_p%d.json'%p)); print('c5 B=64 one GPU pieces',p,'step %.1f us value %.3f M' % (d['ms_per_step']*1e3, d['value']/1e6))
"
tail -3 gpurun_out/r3b/err.txt
