#!/bin/bash
# round 5: A/B of build variants on one box: tools/r05_variants.sh "name:flags" ...   (geometries: ico cad)
run() { MVR_LIB=$2 python bench.py --geometry $3 --steps 300 --warmup 50 --no-pipelined --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $3', 'ms/step %.4f raster %.4f'%(d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  python mvtn_amd/build.py /tmp/var_$name.so $flags > /dev/null 2>&1 || { echo "$name: build failed"; continue; }
  for g in ${GEOS:-ico cad}; do run $name /tmp/var_$name.so $g; done
done
