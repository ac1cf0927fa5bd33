python -m pytest tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r2_t11.log
for tile in 22 42 44; do for wgs in 1536 3072 6144; do echo "### tile $tile wgs $wgs"; TG_TN_TILE=$tile TG_TN_WGS=$wgs python3 tools/tn_tile_lab.py 2>&1 | grep -v amdgpu.ids | head -4; done; done > gpurun_out/r2_i_tn_lab.txt 2>&1
python3 tools/gru_cluster_probe.py > gpurun_out/r2_i_gru_probe.txt 2>&1
bash tools/r2_profile.sh r2_i > gpurun_out/r2_i_profile.log 2>&1
cat gpurun_out/r2_t11.log gpurun_out/r2_i_tn_lab.txt gpurun_out/r2_i_gru_probe.txt; tail -c 300 gpurun_out/r2_i_bench.json; head -16 gpurun_out/r2_i_by_shape.txt
