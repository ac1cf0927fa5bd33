python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r2_t7.log
bash tools/r2_profile.sh r2_e > gpurun_out/r2_e_profile.log 2>&1
python tools/gemm_census.py > gpurun_out/r2_e_census.txt 2>&1
cat gpurun_out/r2_t7.log; tail -c 700 gpurun_out/r2_e_bench.json
