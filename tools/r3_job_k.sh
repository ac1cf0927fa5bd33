#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "mover_wave or gemm_nt" > $out/r3_k_tests.log 2>&1
echo "tests rc=$?"; tail -3 $out/r3_k_tests.log
timeout -k 10 300 python3 tools/nt_mw_probe.py 5 > $out/r3_k_nt_mw_probe.txt 2>&1
grep -v amdgpu.ids $out/r3_k_nt_mw_probe.txt | tail -6
timeout -k 10 200 python3 tools/mw_ablate.py > $out/r3_k_mw_ablate.txt 2>&1
grep -v amdgpu.ids $out/r3_k_mw_ablate.txt
rm -f $out/r3_k_mw_roles.txt
for a in 0 14; do timeout -k 10 100 python3 tools/mw_roles.py $a gru >> $out/r3_k_mw_roles.txt 2>&1; done
grep -v amdgpu.ids $out/r3_k_mw_roles.txt
