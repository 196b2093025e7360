# developer scratch: build variants of the library with -D flags and bench
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "-DMVR_PB_LDS_VERTS=12288" "-DMVR_PB_LDS_VERTS=4096"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math $v -o /tmp/libv.so mvtn_amd/csrc/mvr_kernels.hip
  d=/tmp/sv_$RANDOM; 
  MVR_LIB=/tmp/libv.so rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --tier L --batch 8 --views 20 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  echo "== $v"; python3 -c "
import csv,glob,sys
for r in csv.DictReader(open(glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True)[0])): print('   ', r['Name'][:44], round(float(r['AverageNs'])/1e3,1))" $d | head -6
  MVR_LIB=/tmp/libv.so python bench.py --tier L --batch 8 --views 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('    step', d['ms_per_step'], d['value'])"
done
