set -o pipefail
mkdir -p gpurun_out
bash tools/r2_profile.sh r2_fin3 > gpurun_out/r2_fin3_profile.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2_fin3_bench.json
timeout -k 10 300 python bench.py > gpurun_out/r2_fin3_bench_full.json 2> gpurun_out/r2_fin3_bench_full.err; head -c 300 gpurun_out/r2_fin3_bench_full.json; echo
timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --force-ddp > gpurun_out/r2_fin3_bench_ddp.json 2> gpurun_out/r2_fin3_bench_ddp.err; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2_fin3_bench_ddp.json
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
