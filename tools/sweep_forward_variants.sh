# developer scratch: build variants of the library with -D flags and bench
set -e
cd $GRAFT_REPO_ROOT
for w in 2048 3072 4096 6144; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -DMVR_PTS_LIST_INTS=$w -o /tmp/lib$w.so mvtn_amd/csrc/mvr_kernels.hip
  MVR_LIB=/tmp/lib$w.so python bench.py --workload points_fwd --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('list ints $w', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
