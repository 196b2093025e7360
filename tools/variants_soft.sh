# developer scratch: soft forward / backward kernel time for prebuilt variant libraries tools/var_*.so (C3 shape, K = 4)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do
for lib in tools/libhead.so tools/var_*.so; do
  export MVR_LIB=$lib
  r=$(timeout 300 python3 bench.py --workload soft_fwdbwd --no-cpu-baseline --steps 30 --warmup 5 2>>/tmp/ab.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step %.0f us  forward %.0f  backward %.0f  err %s' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline'].get('backward_kernel_ms',0)*1e3, d.get('max_pixel_err_vs_oracle')))")
  echo "$lib | $r"
done
done
tail -2 /tmp/ab.err
