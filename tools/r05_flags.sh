#!/bin/bash
# round 5: compiler-flag variants of the whole library timed on one box (ico / cad steps): tools/r05_flags.sh "name:flags" ...
run() { MVR_LIB=$2 python3 bench.py --geometry $3 --steps 300 --warmup 50 --no-pipelined --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $3', 'us/step %.1f raster %.1f'%(d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"; }
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  python3 mvtn_amd/build.py /tmp/var_$name.so $flags > /tmp/build_$name.log 2>&1 || { echo "$name: build failed: $(tail -2 /tmp/build_$name.log | head -1 | cut -c1-160)"; continue; }
  for g in ${GEOS:-ico}; do run $name /tmp/var_$name.so $g; done
done
