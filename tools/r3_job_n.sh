#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q --durations=25 -s > $out/r3_n_tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed" $out/r3_n_tests.log | tail -3
grep -E "gates|flipped|per-iteration|optimiser state|worst tensor|graph x5" $out/r3_n_tests.log | head -20
grep -A28 "slowest" $out/r3_n_tests.log | head -32
