#!/bin/bash
# round 6: the "close" corner with stacked soups (2-4 coincident copies: several candidates per pixel) -> gpurun_out/r06_stress_close2.txt
out=gpurun_out/r06_stress_close2.txt; : > $out
for seed in ${SEEDS:-11 12}; do
( echo "== close corner with stacks, seed $seed, ${1:-8000} cases"; timeout 1200 python3 tests/stress_parity.py ${1:-8000} $seed 0 close > /tmp/s_$seed.txt 2>&1
  grep -c "^case" /tmp/s_$seed.txt; grep -B1 -A14 "MISMATCH" /tmp/s_$seed.txt | head -60; grep "DISAGREE" /tmp/s_$seed.txt | head -5; tail -2 /tmp/s_$seed.txt ) >> $out 2>&1
done
cat $out
