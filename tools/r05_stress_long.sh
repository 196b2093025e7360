#!/bin/bash
# round 5: long stress runs on the library as shipped -> gpurun_out/r05_stress_long.txt
out=gpurun_out/r05_stress_long.txt; : > $out
( timeout 3000 python tests/stress_parity.py ${1:-80000} 11 | tail -3 ) >> $out 2>&1
( timeout 1200 python tests/stress_points.py 10000 11 | tail -2 ) >> $out 2>&1
( timeout 1200 python tests/stress_soft.py 1500 11 | tail -2 ) >> $out 2>&1
( timeout 1200 python tests/stress_window.py 2000 11 | tail -2 ) >> $out 2>&1
cat $out
