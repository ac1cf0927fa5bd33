#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
for a in 0 6 2 4 16; do python3 tools/mw_roles.py $a gru >> $out/r3_f_mw_roles.txt 2>&1; done
python3 tools/mw_roles.py 0 tcn >> $out/r3_f_mw_roles.txt 2>&1
python3 tools/mw_roles.py 0 gru0 >> $out/r3_f_mw_roles.txt 2>&1
grep -v amdgpu.ids $out/r3_f_mw_roles.txt
