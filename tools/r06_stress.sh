#!/bin/bash
# round 6: the stress runs against the oracle on the library as shipped -> gpurun_out/r06_stress.txt (tail of each run + its verdict)
out=gpurun_out/r06_stress.txt; : > $out
( timeout 2700 python tests/stress_parity.py ${1:-20000} ${2:-6} | tail -3 ) >> $out 2>&1
( timeout 900 python tests/stress_points.py 3000 6 | tail -2 ) >> $out 2>&1
( timeout 900 python tests/stress_soft.py 300 6 | tail -2 ) >> $out 2>&1
( timeout 900 python tests/stress_window.py 500 6 | tail -2 ) >> $out 2>&1
cat $out
