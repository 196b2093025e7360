#!/bin/bash
# round 6: long stress run on the library as shipped -> gpurun_out/r06_stress_long.txt
out=gpurun_out/r06_stress_long.txt; : > $out
( timeout 4200 python tests/stress_parity.py ${1:-60000} 13 | tail -3 ) >> $out 2>&1
( timeout 900 python tests/stress_soft.py 1500 13 | tail -2 ) >> $out 2>&1
( timeout 900 python tests/stress_points.py 8000 13 | tail -2 ) >> $out 2>&1
cat $out
