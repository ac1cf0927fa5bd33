#!/bin/bash
# round 3, job s: bf16-tier profile (by-shape table), PMC passes, then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
BENCH_ARGS="--dtype bf16" bash tools/r3_profile.sh r3_s_bf16 > gpurun_out/r3_s_profile.log 2>&1 &&
bash tools/r3_pmc.sh > gpurun_out/r3_s_pmc.log 2>&1 &&
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -s --durations=15 > gpurun_out/r3_s_tests.log 2>&1
rc=$?; echo "rc=$rc"; tail -4 gpurun_out/r3_s_tests.log; cat gpurun_out/r3_s_pmc.log
