python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r2_t12.log
for r in 1 2 4; do TG_H64_RING=$r python3 tools/h64_probe.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r2_j_h64_probe.txt
python3 tools/gru_cluster_probe.py 2>&1 | grep "us/step" > gpurun_out/r2_j_gru_probe.txt
bash tools/r2_profile.sh r2_j > gpurun_out/r2_j_profile.log 2>&1
cat gpurun_out/r2_t12.log gpurun_out/r2_j_h64_probe.txt gpurun_out/r2_j_gru_probe.txt; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2_j_bench.json; head -24 gpurun_out/r2_j_by_shape.txt
