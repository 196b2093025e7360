cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r04_e; mkdir -p $out
w=points_fwd
timeout 400 python3 bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$w -- python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${c}_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
done
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq1_points_fwd -- python3 bench.py --workload points_fwd --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
tail -1 $out/bench_$w.json | cut -c1-300
