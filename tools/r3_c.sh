cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3c
run() {  # label, env...
  for w in mesh_fwd c5_per_gpu; do
    timeout 200 python3 bench.py --workload $w --no-cpu-baseline --no-pipelined --steps 1000 --warmup 100 2>>gpurun_out/r3c/err.txt | tail -1 > gpurun_out/r3c/b.json
    python3 -c "
import json
d=json.load(open('gpurun_out/r3c/b.json'))
print('$1 $w step %.1f us  kernel %.1f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
  done
}
MVR_FAST_TABEXT=0 run tab198
MVR_DEBUG_OCC=1 python3 bench.py --workload mesh_fwd --no-cpu-baseline --no-pipelined --steps 5 --warmup 2 2>&1 | grep "workgroups/CU" | head -2
run tab250
MVR_FAST_TABEXT=0 run tab198
run tab250
timeout 300 python3 tools/fast_check.py 2>&1 | tail -4
timeout 600 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
tail -3 gpurun_out/r3c/err.txt
