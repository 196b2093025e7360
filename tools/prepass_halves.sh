# developer scratch: pre-pass time with the vertex-normal work of a mesh split over 1 / 2 / 4 workgroups
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 6400 1400 700; do
  python3 mvtn_amd/build.py /tmp/libpp.so -DMVR_NRM_LDS_VERTS=$v > /dev/null
  d=/tmp/pp_prof_$v; rm -rf $d
  MVR_LIB=/tmp/libpp.so rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== NRM_LDS_VERTS $v: prepass $(grep k_mesh_prepass $f | awk -F, '{print $(NF-4)}')"
done
