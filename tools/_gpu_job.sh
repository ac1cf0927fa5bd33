mkdir -p gpurun_out
python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "golden or graph or odd or dropout or variants" > gpurun_out/s11_engine.log 2>&1; tail -5 gpurun_out/s11_engine.log
bash tools/r2_profile.sh r2_t > gpurun_out/r2_t_profile.log 2>&1; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r2_t_bench.json
grep "speaker\|bn_small" gpurun_out/r2_t_timeline.txt | tail -8
