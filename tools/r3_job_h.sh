#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "planes or gemm_nt or gru_layer or weight_prep" > $out/r3_h_planes_tests.log 2>&1
echo "planes tests rc=$?"; tail -8 $out/r3_h_planes_tests.log
timeout -k 10 300 python3 tools/nt_mw_probe.py 3 > $out/r3_h_nt_mw_probe.txt 2>&1
echo "probe rc=$?"; grep -v amdgpu.ids $out/r3_h_nt_mw_probe.txt | head -8
timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > $out/r3_h_bench.json 2> $out/r3_h_bench.err
echo "bench rc=$?"; head -c 400 $out/r3_h_bench.json; tail -3 $out/r3_h_bench.err
TG_GEMM_PLANES=0 timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > $out/r3_h_bench_noplanes.json 2> $out/r3_h_bench_noplanes.err
echo "bench(no planes) rc=$?"; head -c 400 $out/r3_h_bench_noplanes.json
TG_GEMM_PLANES=0 TG_NT_MW=0 timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > $out/r3_h_bench_old.json 2> $out/r3_h_bench_old.err
echo "bench(old kernels) rc=$?"; head -c 400 $out/r3_h_bench_old.json
