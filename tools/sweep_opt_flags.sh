# Developer sweep (GPU box): optimiser switches for the whole library; prints step / kernel time per workload.
cd $GRAFT_REPO_ROOT
for v in "" "-fno-vectorize" "-fno-unroll-loops" "-O2" "-mllvm -amdgpu-early-inline-all=true" "-fno-strict-aliasing" "-mllvm -amdgpu-atomic-optimizer-strategy=None" "-mllvm -amdgpu-load-store-vectorizer=0" "-mllvm -amdgpu-scalarize-global-loads=0" "-mllvm -amdgpu-enable-vopd=0" "-mllvm -amdgpu-dpp-combine=0"; do
  if python3 mvtn_amd/build.py /tmp/libs.so $v > /tmp/build.log 2>&1; then
    for w in mesh_fwd mesh_fwdbwd points_fwd; do
      MVR_LIB=/tmp/libs.so timeout 120 python3 bench.py --workload $w --no-cpu-baseline --no-pipelined --steps 300 --warmup 100 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$v] $w step %.1f kernel %.1f %s' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, ('bwd %.1f' % (r['backward_kernel_ms']*1e3)) if 'backward_kernel_ms' in r else ''))"
    done
  else
    echo "[$v] build failed: $(tail -1 /tmp/build.log | cut -c1-150)"
  fi
done
