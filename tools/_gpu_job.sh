set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm_nt or grouped or conv" > gpurun_out/s41_ops.log 2>&1; tail -5 gpurun_out/s41_ops.log
timeout -k 10 400 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "golden or graph or variant" > gpurun_out/s41_engine.log 2>&1; tail -3 gpurun_out/s41_engine.log
timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/s41_bench.json 2> gpurun_out/s41_bench.err && grep -o '"ms_per_step": [0-9.]*' gpurun_out/s41_bench.json
TG_NT_EPILOGUE_EXT=0 timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/s41_bench_off.json 2> gpurun_out/s41_bench_off.err && grep -o '"ms_per_step": [0-9.]*' gpurun_out/s41_bench_off.json
