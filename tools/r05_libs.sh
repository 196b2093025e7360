#!/bin/bash
# round 5: prebuilt library variants (tools/*.so travel with the snapshot) timed alternately on one box: tools/r05_libs.sh lib1.so lib2.so ...
for rep in 1 2; do
for lib in "$@"; do
  for w in ${WL:-mesh_fwd}; do
    MVR_LIB=$lib python3 bench.py --workload $w --no-cpu-baseline --no-pipelined --steps 500 --warmup 100 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib $w step %.1f kernel %.1f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
  done
done
done
