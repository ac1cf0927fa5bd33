set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm or grouped or conv or planes" > gpurun_out/s32_ops.log 2>&1; tail -3 gpurun_out/s32_ops.log
TG_NT_FAST=0 timeout -k 10 100 python tools/gemm_probe.py > gpurun_out/s32_probe_fast0.txt 2>&1 && TG_NT_FAST=1 timeout -k 10 100 python tools/gemm_probe.py > gpurun_out/s32_probe_fast1.txt 2>&1 && TG_NT_FAST=1 TG_NT_RING=2 timeout -k 10 100 python tools/gemm_probe.py > gpurun_out/s32_probe_fast1_ring2.txt 2>&1
paste -d'\n' <(grep "^nt" gpurun_out/s32_probe_fast0.txt | cut -c1-72) <(grep "^nt" gpurun_out/s32_probe_fast1.txt | cut -c1-72) <(grep "^nt" gpurun_out/s32_probe_fast1_ring2.txt | cut -c1-72)
timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/s32_bench.json 2> gpurun_out/s32_bench.err && grep -o '"ms_per_step": [0-9.]*' gpurun_out/s32_bench.json
TG_NT_FAST=0 timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/s32_bench_fast0.json 2> gpurun_out/s32_bench_fast0.err && grep -o '"ms_per_step": [0-9.]*' gpurun_out/s32_bench_fast0.json
