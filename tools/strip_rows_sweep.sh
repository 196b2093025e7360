cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for th in 8 10 12 14 16 20; do
  rm -rf /tmp/vp; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-pipelined --strip-rows $th > /tmp/b.json 2>/dev/null
  echo "== TH $th  $(cut -c1-160 /tmp/b.json | grep -o 'ms_per_step[^,]*')"
  find /tmp/vp -name "*kernel_stats.csv" | head -1 | xargs grep "raster\|prepass" | cut -d, -f1-4 | sed 's/(.*)//'
done
