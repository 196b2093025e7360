#!/bin/bash
# round 6, the library as shipped (after the settle's key fix and the backward's gradient form): standard-mode mesh stress on a new seed,
# window / soft / point stress on new seeds -> gpurun_out/r06_stress_final.txt
out=gpurun_out/r06_stress_final.txt; : > $out
( echo "== mesh, seed 97, ${1:-20000} cases"; timeout 1500 python3 tests/stress_parity.py ${1:-20000} 97 > /tmp/sf.txt 2>&1; grep -c "^case" /tmp/sf.txt; grep "MISMATCH\|DISAGREE" /tmp/sf.txt | head -5; tail -2 /tmp/sf.txt ) >> $out 2>&1
( echo "== window, seed 97"; timeout 300 python3 tests/stress_window.py 1000 97 2>&1 | tail -2 ) >> $out 2>&1
( echo "== soft, seed 97"; timeout 400 python3 tests/stress_soft.py 300 97 2>&1 | tail -2 ) >> $out 2>&1
( echo "== points, seed 97"; timeout 300 python3 tests/stress_points.py 3000 97 2>&1 | tail -2 ) >> $out 2>&1
cat $out
