python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r2_t8.log
bash tools/r2_profile.sh r2_f > gpurun_out/r2_f_profile.log 2>&1
for m in plain cap seg; do
  case $m in plain) fl="";; cap) fl="--force-ddp"; export TG_DDP_CAPTURE=1;; seg) fl="--force-ddp"; export TG_DDP_CAPTURE=0;; esac
  python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline $fl 2> gpurun_out/r2_f_ddp_$m.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'])" >> gpurun_out/r2_f_ddp.txt
done
unset TG_DDP_CAPTURE
bash tools/r2_pmc.sh > gpurun_out/r2_pmc.log 2>&1
python3 bench.py > gpurun_out/r2_f_bench_full.json 2> gpurun_out/r2_f_bench_full.err
cat gpurun_out/r2_t8.log gpurun_out/r2_f_ddp.txt; tail -c 1200 gpurun_out/r2_f_bench_full.json
