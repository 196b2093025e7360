#!/bin/bash
# round 6: the longest stress run of the round on the library as shipped (random MVR_PROJECTION_FMA included) -> gpurun_out/r06_stress_xl.txt
out=gpurun_out/r06_stress_xl.txt; : > $out
( timeout 5000 python tests/stress_parity.py ${1:-120000} 31 | tail -3 ) >> $out 2>&1
( timeout 600 python tests/stress_window.py 2000 31 | tail -2 ) >> $out 2>&1
cat $out
