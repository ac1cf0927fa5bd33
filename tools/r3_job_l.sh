#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
rm -f $out/r3_l_mw_roles.txt
for a in 0 14; do timeout -k 10 100 python3 tools/mw_roles.py $a gru >> $out/r3_l_mw_roles.txt 2>&1; done
grep -v amdgpu.ids $out/r3_l_mw_roles.txt
