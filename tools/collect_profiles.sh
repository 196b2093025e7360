# Collects the round's judged measurements on the GPU box into gpurun_out/$1 (run from the repo root through gpurun):
#   bench lines (un-profiled), rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes and SQ counter passes.
#   (c5 = BASELINE configs[4] on one GPU: all 64 objects; c5_per_gpu = one GPU's share at N = 8, the `c5` of rounds 1-2)
# Every profiler run is bounded by `timeout` (a counter combination that stalls must not eat the GPU budget).
out=gpurun_out/${1:-prof}
mkdir -p $out
export TMPDIR=/tmp
for w in mesh_fwd mesh_fwdbwd points_fwd c5_per_gpu c5 soft_fwdbwd; do
  st=""; [ $w = soft_fwdbwd ] && st="--steps 100 --warmup 10 --cpu-views 4"; [ $w = c5 ] && st="--steps 200 --warmup 20"
  timeout 400 python3 bench.py --workload $w $st --no-secondary > $out/bench_$w.json 2> $out/bench_$w.err
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$w -- python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined --no-secondary > /dev/null 2>&1
done
for w in mesh_fwd mesh_fwdbwd points_fwd; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${c}_$w -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  done
done
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/pmc_sq1_mesh_fwdbwd -- python3 bench.py --workload mesh_fwdbwd --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc_sq2_mesh_fwdbwd -- python3 bench.py --workload mesh_fwdbwd --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $out/pmc_sq3_mesh_fwdbwd -- python3 bench.py --workload mesh_fwdbwd --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
# (the forward rasteriser of the INFERENCE step: in the training step above it also writes the pixel lists)
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/pmc_sq1_mesh_fwd -- python3 bench.py --workload mesh_fwd --steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq1_points_fwd -- python3 bench.py --workload points_fwd --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $out/pmc_sq1_soft_fwdbwd -- python3 bench.py --workload soft_fwdbwd --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
# round 5: the C2-shaped forward on the other two geometries of bench.py --geometry (CAD-like assemblies, ragged S / M / L batches)
for g in cad ragged; do
  timeout 400 python3 bench.py --geometry $g --cpu-views 4 > $out/bench_geo_$g.json 2> $out/bench_geo_$g.err
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_geo_$g -- python3 bench.py --geometry $g --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
done
# round 6: the instruction counters of the CAD-like forward (VERDICT r5 judged it without any)
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/pmc_sq1_geo_cad -- python3 bench.py --geometry cad --steps 3 --warmup 2 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
# what the driver runs at the end of the round (its flags; N = 1)
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_like.json 2> $out/bench_driver_like.err
find $out -name "*kernel_trace.csv" -delete
find $out -name "*agent_info.csv" -delete
du -sh $out
for w in mesh_fwd mesh_fwdbwd points_fwd c5_per_gpu c5 soft_fwdbwd geo_cad geo_ragged; do tail -1 $out/bench_$w.json | cut -c1-400; done
