# developer scratch: A/B of a baseline build against the working tree ON ONE BOX (boxes differ by ~2 us).
# HERE, before the gpurun call:   bash tools/ab_head.sh build [REV]   -> tools/libhead.so from `git archive REV` (default HEAD),
# built under /tmp (no source copy in the tree; the .so is git-ignored and travels with the snapshot).
# ON THE BOX:                      bash tools/ab_head.sh
if [ "$1" = build ]; then
  rev=${2:-HEAD}; rm -rf /tmp/mvr_head && mkdir -p /tmp/mvr_head && git archive $rev mvtn_amd include | tar -x -C /tmp/mvr_head &&
  python3 /tmp/mvr_head/mvtn_amd/build.py $(pwd)/tools/libhead.so > /dev/null && echo "tools/libhead.so <- $rev"
  exit
fi
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2 3; do
for lib in tools/libhead.so mvtn_amd/libmvr_hip.so; do
  export MVR_LIB=$lib
  for w in mesh_fwd c5_per_gpu; do
    timeout 100 python3 bench.py --workload $w --no-cpu-baseline --no-pipelined --steps 500 --warmup 100 2>>/tmp/ab.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib $w step %.1f kernel %.1f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
  done
  [ "$AB_FWDBWD" = 1 ] && timeout 100 python3 bench.py --workload mesh_fwdbwd --no-cpu-baseline --steps 300 --warmup 50 2>>/tmp/ab.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib fwdbwd eager %.1f graph %.1f raster %.1f bwd %.1f' % (d['ms_per_step']*1e3, d['hip_graph_replay']['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline']['backward_kernel_ms']*1e3))"
done
done
tail -3 /tmp/ab.err
