cd $GRAFT_REPO_ROOT
for w in 2 3 4; do
  python3 mvtn_amd/build.py /tmp/libb$w.so -DMVR_BWD_STRIP_MULT=4 -DMVR_BWD_WAVES_PER_SIMD=$w > /dev/null
  MVR_LIB=/tmp/libb$w.so python bench.py --workload mesh_fwdbwd --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bwd waves $w', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['backward_kernel_ms'])"
done
