#!/bin/bash
# round 6: a second long stress run on the library as shipped, other seed -> gpurun_out/r06_stress_xl2.txt
out=gpurun_out/r06_stress_xl2.txt; : > $out
( timeout 5500 python tests/stress_parity.py ${1:-130000} 41 | tail -3 ) >> $out 2>&1
cat $out
