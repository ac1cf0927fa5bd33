python -m pytest tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -3 > gpurun_out/r2_t14.log
python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "sticky or graphed or parallel or odd or oracle" 2>&1 | tail -3 >> gpurun_out/r2_t14.log
timeout 200 python3 tools/gru_cluster_soak.py 90 > gpurun_out/r2_l_soak.txt 2>&1
bash tools/r2_profile.sh r2_l > gpurun_out/r2_l_profile.log 2>&1
cat gpurun_out/r2_t14.log; tail -3 gpurun_out/r2_l_soak.txt; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2_l_bench.json; head -30 gpurun_out/r2_l_by_shape.txt; grep -c . gpurun_out/r2_l_timeline.txt
