#!/bin/bash
# A/B of prebuilt libraries on ONE box, alternately (tools/*.so travel with the snapshot):
#   CASES="ico cad ragged c5_per_gpu c3 c4" REPS=2 tools/ab_libs.sh lib1.so lib2.so ...
# prints per (library, case): step us, dominant kernel us, pre-pass us
args_of() { case $1 in
  ico|cad|ragged) echo "--geometry $1";;
  c5_per_gpu) echo "--workload c5_per_gpu";;
  c3) echo "--workload mesh_fwdbwd";;
  c4) echo "--workload points_fwd";;
  soft) echo "--workload soft_fwdbwd --steps 60 --warmup 10";;
esac; }
for rep in $(seq 1 ${REPS:-2}); do
for lib in "$@"; do
  for c in ${CASES:-ico cad}; do
    MVR_LIB=$lib python3 bench.py $(args_of $c) --steps 300 --warmup 50 --no-cpu-baseline --no-pipelined --no-secondary $(args_of $c | grep -q steps && echo "" ) 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; fs=d.get('fast_raster_stats') or {}
print('%-40s %-10s step %7.1f  kernel %7.1f  bwd %s  prepass %s' % ('$lib', '$c', d['ms_per_step']*1e3, r['kernel_ms']*1e3, ('%.1f' % (r['backward_kernel_ms']*1e3)) if 'backward_kernel_ms' in r else '-', ('%.1f' % (fs['prepass_kernel_ms']*1e3)) if fs.get('prepass_kernel_ms') else '-'))"
  done
done
done
