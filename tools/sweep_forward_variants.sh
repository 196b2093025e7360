# developer scratch: build variants of the library with -D flags and bench
set -e
cd $GRAFT_REPO_ROOT
for v in "-DMVR_RB=256" "-DMVR_RB=512" "-DMVR_RB=128"; do
  python3 mvtn_amd/build.py /tmp/libv.so $v > /dev/null
  MVR_LIB=/tmp/libv.so python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
