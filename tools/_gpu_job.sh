set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "tn or grouped or conv" > gpurun_out/s40_ops.log 2>&1; tail -3 gpurun_out/s40_ops.log
for cfg in "TG_TN_WGS22=2000" "TG_TN_WGS22=2900" "TG_TN_WGS22=1500"; do
  env $cfg timeout -k 10 200 python bench.py --steps 150 --warmup 30 --no-cpu-baseline > gpurun_out/s40_bench.json 2> gpurun_out/s40_bench.err && echo "$cfg $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/s40_bench.json)"
done
