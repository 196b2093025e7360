#!/bin/bash
# round 5: the C2-shaped forward on the three geometries of bench.py --geometry; JSON lines -> gpurun_out/r05_geo_<tag>_<geometry>.json
tag=${1:-a}
for g in ico cad ragged; do
  python bench.py --geometry $g --steps 300 --warmup 50 --no-pipelined --cpu-views 4 > gpurun_out/r05_geo_${tag}_$g.json 2> gpurun_out/r05_geo_${tag}_$g.err
  python - "$g" gpurun_out/r05_geo_${tag}_$g.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], "views/s %.0f  ms/step %.4f  raster ms %.4f  stats %s  parity %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["fast_raster_stats"], d["max_pixel_err_vs_oracle"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
