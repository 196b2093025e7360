cd $GRAFT_REPO_ROOT
for v in "" "-DMVR_ABL_PTS_NO_STORE" "-DMVR_ABL_PTS_NO_SCAN" "-DMVR_ABL_PTS_NO_SCAN -DMVR_ABL_PTS_NO_STORE"; do
  python3 mvtn_amd/build.py /tmp/libp.so $v > /dev/null
  MVR_LIB=/tmp/libp.so python bench.py --workload points_fwd --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
