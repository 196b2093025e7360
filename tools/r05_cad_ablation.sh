#!/bin/bash
# round 5: where k_mesh_raster_fast's time goes on the CAD-like batch (ablation builds; results of the ablated builds are wrong)
run() { MVR_LIB=$2 python bench.py --geometry ${3:-cad} --steps 100 --warmup 20 --no-pipelined --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'ms/step %.4f raster %.4f'%(d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for v in "full:" "no_wide:-DMVR_ABL_FAST_NO_WIDE" "no_settle:-DMVR_ABL_FAST_NO_SETTLE" "no_wide_no_settle:-DMVR_ABL_FAST_NO_WIDE -DMVR_ABL_FAST_NO_SETTLE" "no_items:-DMVR_ABL_FAST_NO_ITEMS" "no_items_no_wide:-DMVR_ABL_FAST_NO_ITEMS -DMVR_ABL_FAST_NO_WIDE -DMVR_ABL_FAST_NO_SETTLE" "$@"; do
  name=${v%%:*}; flags=${v#*:}
  python mvtn_amd/build.py /tmp/abl_$name.so $flags > /dev/null 2>&1
  run $name /tmp/abl_$name.so cad
done
