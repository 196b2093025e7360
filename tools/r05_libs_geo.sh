#!/bin/bash
# round 5: prebuilt libraries (tools/*.so travel with the snapshot) timed alternately on one box, per geometry: GEOS="ico cad" tools/r05_libs_geo.sh lib1.so lib2.so ...
for rep in 1 2; do
for lib in "$@"; do
  for g in ${GEOS:-ico cad}; do
    MVR_LIB=$lib python3 bench.py --geometry $g --no-cpu-baseline --no-pipelined --steps 300 --warmup 50 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib $g step %.1f raster %.1f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
  done
done
done
