# developer scratch: dynamic instruction counts of k_mesh_soft_forward via ablation builds (C3 shape)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in full "-DMVR_ABL_SOFT_NO_BLEND" "-DMVR_ABL_SOFT_NO_EVAL -DMVR_ABL_SOFT_NO_BLEND"; do
  d=""; [ "$v" != full ] && d="$v"
  python3 mvtn_amd/build.py /tmp/libabl.so $d > /dev/null
  export MVR_LIB=/tmp/libabl.so
  rm -rf /tmp/pmc_abl
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_abl -- python3 bench.py --workload soft_fwdbwd --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== $v"; python3 tools/pmc_summary.py /tmp/pmc_abl k_mesh_soft_forward | tr -d '\n '; echo
done
