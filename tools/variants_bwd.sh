# developer scratch: C3 step (HIP-graph replay) and backward-kernel time for prebuilt variant libraries tools/var_*.so, three
# interleaved repetitions on ONE box (the forward's tools/variants.sh for the training step)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2 3; do
for lib in tools/libhead.so tools/var_*.so; do
  export MVR_LIB=$lib
  r=$(timeout 200 python3 bench.py --workload mesh_fwdbwd --no-cpu-baseline --steps 300 --warmup 50 2>>/tmp/ab.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); g=d.get('hip_graph_replay') or {}; print('eager %.1f graph %.1f raster %.1f backward %.1f' % (d['ms_per_step']*1e3, g.get('ms_per_step',0)*1e3, d['roofline']['kernel_ms']*1e3, d['roofline'].get('backward_kernel_ms',0)*1e3))")
  echo "$lib | $r"
done
done
tail -2 /tmp/ab.err
