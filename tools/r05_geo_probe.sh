run() { python bench.py --geometry cad --steps 100 --warmup 20 --no-pipelined --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'ms/step %.4f raster %.4f'%(d['ms_per_step'], d['roofline']['kernel_ms']), d['fast_raster_stats'])"; }
run default
MVR_RASTER_MODE=exact run exact_only
MVR_CAD_DOUBLE_SIDED=0 run no_doubles
MVR_CAD_DOUBLE_SIDED=0 MVR_RASTER_MODE=exact run no_doubles_exact
