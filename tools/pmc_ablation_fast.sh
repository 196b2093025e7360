# developer scratch: dynamic VALU instruction count per phase of k_mesh_raster_fast via ablation builds
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in full "-DMVR_ABL_FAST_NO_ITEMS" "-DMVR_ABL_FAST_NO_ITEMS -DMVR_ABL_FAST_NO_RESOLVE" "-DMVR_ABL_FAST_RETURN_AT=2" "-DMVR_ABL_FAST_RETURN_AT=1" "-DMVR_ABL_FAST_RETURN_AT=0" "-DMVR_ABL_NO_SHADE"; do
  d=""; [ "$v" != full ] && d="$v"
  python3 mvtn_amd/build.py /tmp/libabl.so $d > /dev/null
  export MVR_LIB=/tmp/libabl.so
  rm -rf /tmp/pmc_abl
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d /tmp/pmc_abl -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
  echo "== $v"; python3 tools/pmc_summary.py /tmp/pmc_abl k_mesh_raster_fast | tr -d '\n '; echo
  rm -rf /tmp/pmc_abl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_abl -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
  find /tmp/pmc_abl -name "*kernel_stats.csv" | head -1 | xargs grep "raster_fast" | cut -d, -f1-4
done
