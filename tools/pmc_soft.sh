# developer scratch: SQ counters of the soft kernels (C3 shape)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/pmc_soft
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d /tmp/pmc_soft -- python3 bench.py --workload soft_fwdbwd --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_summary.py /tmp/pmc_soft k_mesh_soft_forward | tr -d '\n '; echo
python3 tools/pmc_summary.py /tmp/pmc_soft k_mesh_soft_backward | tr -d '\n '; echo
