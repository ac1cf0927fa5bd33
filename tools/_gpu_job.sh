set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x > gpurun_out/s31_engine.log 2>&1; tail -3 gpurun_out/s31_engine.log
timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/s31_bench.json 2> gpurun_out/s31_bench.err && grep -o '"ms_per_step": [0-9.]*' gpurun_out/s31_bench.json
TG_D_DEFER_WGRAD=0 timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/s31_bench_nodefer.json 2> gpurun_out/s31_bench_nodefer.err && grep -o '"ms_per_step": [0-9.]*' gpurun_out/s31_bench_nodefer.json
TG_TN_WGS=1536 timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/s31_bench_1536.json 2> gpurun_out/s31_bench_1536.err && grep -o '"ms_per_step": [0-9.]*' gpurun_out/s31_bench_1536.json
