cd $GRAFT_REPO_ROOT
bash tools/variants.sh
export TMPDIR=/tmp
for rep in 1 2; do
for own in 1 0; do
  MVR_FAST_PLIST_OWN=$own MVR_LIB=tools/var_a_new.so python3 bench.py --workload mesh_fwd --no-cpu-baseline --no-pipelined --steps 500 --warmup 100 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plist_own=$own step %.1f kernel %.1f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
done
done
MVR_LIB=tools/var_a_new.so timeout 200 python3 tools/fast_check.py 2>&1 | tail -1
