mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "residual or elementwise" > gpurun_out/s15_ops.log 2>&1; tail -3 gpurun_out/s15_ops.log
python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "golden or graph" > gpurun_out/s15_engine.log 2>&1; tail -3 gpurun_out/s15_engine.log
bash tools/r2_profile.sh r2_v > gpurun_out/r2_v_profile.log 2>&1; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r2_v_bench.json
grep "act_mask\|add_relu\|mul_k" gpurun_out/r2_v_timeline.txt | tail -4
