# developer scratch: step / raster-kernel time of the mesh forward for prebuilt variant libraries tools/var_*.so (C2 and the
# C5 share), three interleaved repetitions on ONE box.  Build them here first:  python3 mvtn_amd/build.py tools/var_NAME.so -D...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2 3; do
for lib in tools/libhead.so tools/var_*.so; do
  export MVR_LIB=$lib
  out="$lib"
  for w in ${VAR_WORKLOADS:-mesh_fwd c5_per_gpu}; do
    r=$(timeout 100 python3 bench.py --workload $w --no-cpu-baseline --no-pipelined --steps 500 --warmup 100 2>>/tmp/ab.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%s step %.1f kernel %.1f' % ('$w', d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))")
    out="$out | $r"
  done
  echo "$out"
done
done
if [ "$VAR_CHECK" = 1 ]; then for lib in tools/var_*.so; do MVR_LIB=$lib timeout 200 python3 tools/fast_check.py 2>&1 | tail -1; done; fi
tail -2 /tmp/ab.err
