python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r2_t6.log
python tools/gru_cluster_probe.py > gpurun_out/r2_gru_probe_x3.txt 2>&1
TG_GRU_X3=0 python tools/gru_cluster_probe.py > gpurun_out/r2_gru_probe_f32.txt 2>&1
rm -f gpurun_out/r2_ab3.txt
for i in 1 2; do
TG_GRU_X3=0 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gru_x3=0', d['ms_per_step'])" >> gpurun_out/r2_ab3.txt
TG_GRU_X3=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gru_x3=1', d['ms_per_step'])" >> gpurun_out/r2_ab3.txt
done
bash tools/r2_profile.sh r2_d > gpurun_out/r2_d_profile.log 2>&1
cat gpurun_out/r2_t6.log gpurun_out/r2_ab3.txt; tail -12 gpurun_out/r2_gru_probe_x3.txt; tail -6 gpurun_out/r2_gru_probe_f32.txt
