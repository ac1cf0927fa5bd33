#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/r3_m_tests.log 2>&1
echo "tests rc=$?"; tail -5 $out/r3_m_tests.log
timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > $out/r3_m_bench.json 2> $out/r3_m_bench.err
echo "bench rc=$?"; head -c 300 $out/r3_m_bench.json; echo; tail -2 $out/r3_m_bench.err
TG_GEMM_PLANES=0 timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > $out/r3_m_bench_noplanes.json 2> $out/r3_m_bench_noplanes.err
echo "bench(no weight planes) rc=$?"; head -c 300 $out/r3_m_bench_noplanes.json; echo
