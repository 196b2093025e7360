#!/bin/bash
# round 5: round-4 kernels (tools/lib_r04_final.so, built from a4a27d0 + a stats stub) vs the working tree, same box, alternating
for rep in 1 2 3; do
for lib in tools/lib_r04_final.so mvtn_amd/libmvr_hip.so; do
  for w in ${WL:-mesh_fwd c5_per_gpu}; do
    MVR_LIB=$lib python3 bench.py --workload $w --no-cpu-baseline --no-pipelined --steps 500 --warmup 100 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib $w step %.1f kernel %.1f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
  done
done
done
