#!/bin/bash
# round-3 first GPU call: the new tests, the data-parallel self-check at world size 1, a baseline profile of the unchanged kernels
set -u
out=gpurun_out; mkdir -p $out
python3 -m pytest tests/test_trajectory_gpu.py tests/test_engine_gpu.py -m gpu -x -q -k "trajectory or five or consecutive or replayed or graphed" -s > $out/r3_a_newtests.log 2>&1
echo "newtests rc=$?"; tail -15 $out/r3_a_newtests.log
python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --force-ddp > $out/r3_a_bench_ddp.json 2> $out/r3_a_bench_ddp.err
echo "ddp rc=$?"; tail -c 600 $out/r3_a_bench_ddp.json; tail -5 $out/r3_a_bench_ddp.err
TG_DDP_CAPTURE=0 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --force-ddp > $out/r3_a_bench_ddp_seg.json 2> $out/r3_a_bench_ddp_seg.err
echo "ddp-seg rc=$?"; tail -c 300 $out/r3_a_bench_ddp_seg.json
bash tools/r2_profile.sh r3_a
