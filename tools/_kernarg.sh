cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in unset 0 1; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  r=$(python3 bench.py --workload mesh_fwd --no-cpu-baseline --no-pipelined --steps 500 --warmup 100 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step %.1f kernel %.1f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))")
  echo "HIP_FORCE_DEV_KERNARG=$v $r"
done
done
