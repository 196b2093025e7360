#!/bin/bash
# round 5: build variants timed on the soft forward + backward step (C3 shape, K = 4): tools/r05_soft_variants.sh "name:flags" ...
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  python3 mvtn_amd/build.py /tmp/var_$name.so $flags > /dev/null 2>&1 || { echo "$name: build failed"; continue; }
  MVR_LIB=/tmp/var_$name.so python3 bench.py --workload soft_fwdbwd --steps 60 --warmup 10 --no-cpu-baseline --no-pipelined 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$name step %.1f us forward %.1f backward %.1f' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, (r.get('backward_kernel_ms') or 0)*1e3))"
done
