cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in full NO_NORMALS NO_VIEWS; do
  d=""; [ "$v" != full ] && d="-DMVR_ABL_PREPASS_$v"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math $d -o /tmp/libpp.so mvtn_amd/csrc/mvr_kernels.hip
  rm -rf /tmp/pp_prof
  MVR_LIB=/tmp/libpp.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_prof -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  f=$(find /tmp/pp_prof -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(grep k_mesh_prepass $f | awk -F, '{print $(NF-4)}')"
done
