#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "planes" > $out/r3_g_planes_tests.log 2>&1
echo "planes tests rc=$?"; tail -8 $out/r3_g_planes_tests.log
timeout -k 10 300 python3 tools/nt_mw_probe.py 5 > $out/r3_g_nt_mw_probe.txt 2>&1
echo "probe rc=$?"; grep -v amdgpu.ids $out/r3_g_nt_mw_probe.txt
