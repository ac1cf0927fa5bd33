set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "golden or graph or discrim" > gpurun_out/s35_engine.log 2>&1; tail -3 gpurun_out/s35_engine.log
timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/s35_bench.json 2> gpurun_out/s35_bench.err && grep -o '"ms_per_step": [0-9.]*' gpurun_out/s35_bench.json
TG_H64_MOVERS=0 timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/s35_bench_off.json 2> gpurun_out/s35_bench_off.err && grep -o '"ms_per_step": [0-9.]*' gpurun_out/s35_bench_off.json
