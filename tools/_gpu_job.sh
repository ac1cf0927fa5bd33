set -o pipefail
mkdir -p gpurun_out
for cfg in "TG_TN_TILE=0" "TG_TN_TILE=42" "TG_TN_TILE=42 TG_TN_WGS=1536" "TG_TN_WGS=2304" "TG_TN_WGS=4096"; do
  env $cfg timeout -k 10 200 python bench.py --steps 150 --warmup 30 --no-cpu-baseline > gpurun_out/s38_bench.json 2> gpurun_out/s38_bench.err && echo "$cfg $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/s38_bench.json)"
done
