#!/bin/bash
# round 6, after the fix of case 61760: (a) seed 41 again around the case (58 000 .. 66 000; the cases before are only drawn),
# (b) the "close" corner on two other seeds -> gpurun_out/r06_stress_fix.txt
out=gpurun_out/r06_stress_fix.txt; : > $out
( echo "== seed 41, cases 58000..65999"; timeout 900 python3 tests/stress_parity.py 66000 41 58000 | grep -v "backward table" | tail -2 ) >> $out 2>&1
for seed in 7 8; do
( echo "== close corner, seed $seed, ${1:-12000} cases"; timeout 900 python3 tests/stress_parity.py ${1:-12000} $seed 0 close | grep "MISMATCH\|DISAGREE\|cases with" | tail -5 ) >> $out 2>&1
done
cat $out
