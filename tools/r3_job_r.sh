#!/bin/bash
# round 3, job r: bf16 tier tests at B = 128 + FGD, the kernels the tier touches, then bf16 and fp32 bench lines back to back
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q -m gpu -s -k "bf16 or gru or h64 or gemm_tn" > gpurun_out/r3_r_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r3_r_tests.log; grep -a "bf16 tier\|evaluate_testset," gpurun_out/r3_r_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 bench.py --dtype bf16 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r3_r_bench_bf16.json 2> gpurun_out/r3_r_bench_bf16.err && tail -1 gpurun_out/r3_r_bench_bf16.json &&
timeout -k 10 200 python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r3_r_bench_f32.json 2> gpurun_out/r3_r_bench_f32.err && tail -1 gpurun_out/r3_r_bench_f32.json
