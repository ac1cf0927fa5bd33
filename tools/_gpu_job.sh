mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/s16_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s16_tests.log; tail -4 gpurun_out/s16_tests.log
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/s16_bench.json 2> gpurun_out/s16_bench.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/s16_bench.json
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --force-ddp > gpurun_out/s16_bench_ddp.json 2> gpurun_out/s16_bench_ddp.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/s16_bench_ddp.json
TG_DDP_CAPTURE=0 python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --force-ddp > gpurun_out/s16_bench_ddp_seg.json 2> gpurun_out/s16_bench_ddp.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/s16_bench_ddp_seg.json
