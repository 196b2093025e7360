cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for th in 8 4 6 12 16 24 32; do
  timeout 200 python3 bench.py --workload soft_fwdbwd --strip-rows $th --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('TH $th step %.3f ms fwd %.3f bwd %.3f err %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('backward_kernel_ms',0), d.get('max_pixel_err_vs_oracle')))"
done
