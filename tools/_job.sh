#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3 4; do
timeout -k 10 300 python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --force-ddp > gpurun_out/r3_bc_bench_ddp$i.json 2> gpurun_out/r3_bc_ddp$i.err; echo "run $i rc=$?"; tail -1 gpurun_out/r3_bc_bench_ddp$i.json | cut -c1-200; grep -a "ddp self-check" gpurun_out/r3_bc_ddp$i.err | tail -2
done
timeout -k 10 300 python3 -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "data_parallel" 2>&1 | tail -2
