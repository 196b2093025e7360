# developer scratch: A/B of the committed kernels against the working tree ON ONE BOX (boxes differ by ~2 us).  Before the
# gpurun call, here:  rm -rf tools/_head_src && mkdir tools/_head_src && git archive HEAD mvtn_amd/csrc include | tar -x -C tools/_head_src
# (tools/_head_src is git-ignored; it travels to the box with the snapshot)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
H=tools/_head_src/mvtn_amd/csrc; g++ -O3 -std=c++17 -fPIC -fvisibility-inlines-hidden -c -o /tmp/pack.o $H/mvr_pack.cpp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -o /tmp/libhead.so $H/mvr_kernels.hip $H/mvr_regularize.hip -Wl,/tmp/pack.o || echo head build failed
for rep in 1 2 3; do
for lib in /tmp/libhead.so mvtn_amd/libmvr_hip.so; do
  export MVR_LIB=$lib
  for w in mesh_fwd c5; do
    timeout 100 python3 bench.py --workload $w --no-cpu-baseline --no-pipelined --steps 500 --warmup 100 2>>/tmp/ab.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib $w step %.1f kernel %.1f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
  done
  timeout 100 python3 bench.py --workload mesh_fwdbwd --no-cpu-baseline --steps 300 --warmup 50 2>>/tmp/ab.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib fwdbwd graph step %.1f raster kernel %.1f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
done
done
tail -3 /tmp/ab.err
