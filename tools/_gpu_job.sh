mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -m gpu -q -x > gpurun_out/s7_ops.log 2>&1; tail -5 gpurun_out/s7_ops.log
python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "golden or graph or fgd or autoencoder or feeder or checkpoint" > gpurun_out/s7_engine.log 2>&1; tail -3 gpurun_out/s7_engine.log
bash tools/r2_profile.sh r2_q > gpurun_out/r2_q_profile.log 2>&1; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r2_q_bench.json
grep "wav_\|permute3\|dgrad_pack" gpurun_out/r2_q_timeline.txt
