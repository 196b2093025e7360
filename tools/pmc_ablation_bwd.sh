# developer scratch: dynamic instruction counts of k_mesh_backward_tab via ablation builds (C3)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in full "-DMVR_ABL_BT_NO_PIXELS" ${BT_EXTRA}; do
  d=""; [ "$v" != full ] && d="$v"
  python3 mvtn_amd/build.py /tmp/libabl.so $d > /dev/null
  export MVR_LIB=/tmp/libabl.so
  rm -rf /tmp/pmc_abl
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc_abl -- python3 bench.py --workload mesh_fwdbwd --steps 3 --warmup 2 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
  echo "== $v"; python3 tools/pmc_summary.py /tmp/pmc_abl k_mesh_backward_tab | tr -d '\n '; echo
  rm -rf /tmp/pmc_abl
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_abl -- python3 bench.py --workload mesh_fwdbwd --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
  find /tmp/pmc_abl -name "*kernel_stats.csv" | head -1 | xargs grep "backward_tab" | cut -d, -f1-4
done
