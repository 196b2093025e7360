#!/bin/bash
# round 6, last run on the library as shipped: standard mix 30 000 cases (seed 123) + the close corner 2 x 10 000 (seeds 21, 22)
out=gpurun_out/r06_stress_last.txt; : > $out
( echo "== mesh, standard mix, seed 123, 30000 cases"; timeout 1500 python3 tests/stress_parity.py 30000 123 > /tmp/sl.txt 2>&1; grep -c "^case" /tmp/sl.txt; grep -B1 -A14 "MISMATCH" /tmp/sl.txt | head -40; grep "DISAGREE" /tmp/sl.txt | head -5; tail -2 /tmp/sl.txt ) >> $out 2>&1
for seed in 21 22; do
( echo "== close corner (with stacks), seed $seed, 10000 cases"; timeout 900 python3 tests/stress_parity.py 10000 $seed 0 close > /tmp/sl_$seed.txt 2>&1; grep -c "^case" /tmp/sl_$seed.txt; grep -B1 -A14 "MISMATCH" /tmp/sl_$seed.txt | head -40; tail -2 /tmp/sl_$seed.txt ) >> $out 2>&1
done
cat $out
