python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r2_t5.log
rm -f gpurun_out/r2_ab2.txt
for i in 1 2; do
TG_GEMM_X3=0 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('x3=0', d['ms_per_step'])" >> gpurun_out/r2_ab2.txt
TG_GEMM_X3=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('x3=1', d['ms_per_step'])" >> gpurun_out/r2_ab2.txt
done
bash tools/r2_profile.sh r2_c > gpurun_out/r2_c_profile.log 2>&1
cat gpurun_out/r2_t5.log gpurun_out/r2_ab2.txt
