cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for pad in 0 8000 20000 45000; do
  export MVR_FAST_LDS_PAD=$pad
  MVR_DEBUG_OCC=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined 2>&1 | grep "\[mvr\]" | head -1
  rm -rf /tmp/vp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
  find /tmp/vp -name "*kernel_stats.csv" | head -1 | xargs grep "raster_fast" | cut -d, -f1-4
done
