#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
for a in 0 31 6 9; do python3 tools/mw_timeline.py $a > $out/r3_d_mw_timeline_$a.txt 2>&1; echo "rc=$?"; cat $out/r3_d_mw_timeline_$a.txt; done
