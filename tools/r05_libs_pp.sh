#!/bin/bash
# as r05_libs.sh, printing the pre-pass kernel's duration too (bench.py's fast_raster_stats.prepass_kernel_ms)
for rep in 1 2; do
for lib in "$@"; do
  for w in ${WL:-mesh_fwd c5_per_gpu}; do
    MVR_LIB=$lib python3 bench.py --workload $w --no-cpu-baseline --no-pipelined --steps 500 --warmup 100 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib $w step %.1f raster %.1f prepass %.1f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, (d['fast_raster_stats'] or {}).get('prepass_kernel_ms', 0)*1e3))"
  done
done
done
