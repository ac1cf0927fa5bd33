mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "wav_conv2" > gpurun_out/s14_ops.log 2>&1; tail -3 gpurun_out/s14_ops.log
bash tools/r2_profile.sh r2_u > gpurun_out/r2_u_profile.log 2>&1; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r2_u_bench.json
grep "wgrad" gpurun_out/r2_u_timeline.txt | tail -3
