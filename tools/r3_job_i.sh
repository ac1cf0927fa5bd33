#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "mover_wave or planes or gemm_nt" > $out/r3_i_tests.log 2>&1
echo "tests rc=$?"; tail -8 $out/r3_i_tests.log
timeout -k 10 300 python3 tools/nt_mw_probe.py 5 > $out/r3_i_nt_mw_probe.txt 2>&1
echo "probe rc=$?"; grep -v amdgpu.ids $out/r3_i_nt_mw_probe.txt | tail -8
timeout -k 10 200 python3 tools/mw_ablate.py > $out/r3_i_mw_ablate.txt 2>&1
echo "ablate rc=$?"; grep -v amdgpu.ids $out/r3_i_mw_ablate.txt
for a in 0 14 8; do timeout -k 10 100 python3 tools/mw_roles.py $a gru >> $out/r3_i_mw_roles.txt 2>&1; done
grep -v amdgpu.ids $out/r3_i_mw_roles.txt
