#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm_tn or grouped or conv_forward or linear_bwd" > $out/r3_o_tn_tests.log 2>&1
echo "tn tests rc=$?"; tail -8 $out/r3_o_tn_tests.log
timeout -k 10 200 python3 tools/tn_mw_probe.py > $out/r3_o_tn_probe.txt 2>&1
TG_TN_MW=0 timeout -k 10 200 python3 tools/tn_mw_probe.py >> $out/r3_o_tn_probe.txt 2>&1
grep -v amdgpu.ids $out/r3_o_tn_probe.txt
timeout -k 10 300 python3 -m pytest tests/test_trajectory_gpu.py -m gpu -x -q > $out/r3_o_traj.log 2>&1
echo "traj rc=$?"; tail -3 $out/r3_o_traj.log
