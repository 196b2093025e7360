# developer scratch: dynamic instruction counts per phase of k_points_raster_shade via ablation builds (C4)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in full "-DMVR_ABL_PTS_NO_SCAN" "-DMVR_ABL_PTS_NO_STORE" ${PTS_EXTRA}; do
  d=""; [ "$v" != full ] && d="$v"
  python3 mvtn_amd/build.py /tmp/libabl.so $d > /dev/null
  export MVR_LIB=/tmp/libabl.so
  rm -rf /tmp/pmc_abl
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc_abl -- python3 bench.py --workload points_fwd --steps 3 --warmup 2 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
  echo "== $v"; python3 tools/pmc_summary.py /tmp/pmc_abl k_points_raster_shade | tr -d '\n '; echo
done
