mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "batchnorm or wav_front" > gpurun_out/s10_ops.log 2>&1; tail -8 gpurun_out/s10_ops.log
python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "golden or graph or fgd or autoencoder or odd" > gpurun_out/s10_engine.log 2>&1; tail -5 gpurun_out/s10_engine.log
bash tools/r2_profile.sh r2_s > gpurun_out/r2_s_profile.log 2>&1; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r2_s_bench.json
grep "bn" gpurun_out/r2_s_timeline.txt | tail -12
