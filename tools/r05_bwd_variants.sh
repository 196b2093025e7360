#!/bin/bash
# round 5: build variants of k_mesh_backward_tab timed on one box (C3 step): tools/r05_bwd_variants.sh "name:flags" ...
run() { MVR_LIB=$2 python3 bench.py --workload mesh_fwdbwd --steps 300 --warmup 50 --no-pipelined --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'us/step %.1f raster %.1f backward %.1f'%(d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, (d['roofline'].get('backward_kernel_ms') or 0)*1e3))"; }
for rep in 1 2; do
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  [ -f /tmp/var_$name.so ] || python3 mvtn_amd/build.py /tmp/var_$name.so $flags > /dev/null 2>&1 || { echo "$name: build failed"; continue; }
  run $name /tmp/var_$name.so
done
done
