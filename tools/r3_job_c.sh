#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
python3 tools/mw_ablate.py > $out/r3_c_mw_ablate.txt 2>&1
echo "ablate rc=$?"; cat $out/r3_c_mw_ablate.txt
python3 -m pytest tests/test_trajectory_gpu.py -m gpu -q -s > $out/r3_c_traj.log 2>&1
echo "traj rc=$?"; grep -E "^  |exp_avg|per-iteration|passed|failed|graph x5" $out/r3_c_traj.log | head -40
