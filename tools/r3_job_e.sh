#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
python3 -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "mover_wave or gemm_nt" > $out/r3_e_mwtests.log 2>&1
echo "mw tests rc=$?"; tail -5 $out/r3_e_mwtests.log
python3 tools/nt_mw_probe.py 7 > $out/r3_e_nt_mw_probe.txt 2>&1
echo "probe rc=$?"; cat $out/r3_e_nt_mw_probe.txt
python3 tools/mw_ablate.py > $out/r3_e_mw_ablate.txt 2>&1
echo "ablate rc=$?"; cat $out/r3_e_mw_ablate.txt
python3 -m pytest tests/test_trajectory_gpu.py -m gpu -q -s > $out/r3_e_traj.log 2>&1
echo "traj rc=$?"; grep -E "^  |exp_avg|per-iteration|passed|failed|graph x5|optimiser" $out/r3_e_traj.log | head -40
