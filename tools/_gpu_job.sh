./tools/gemm_split_lab.bin > gpurun_out/r2_split_lab2.txt 2>&1
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gru" 2>&1 | tail -8 > gpurun_out/r2_t4.log
python tools/x3_parity_probe.py > gpurun_out/r2_probe_x3.txt 2>&1
TG_GEMM_X3=0 python tools/x3_parity_probe.py > gpurun_out/r2_probe_f32.txt 2>&1
for i in 1 2; do
TG_GEMM_X3=0 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('x3=0', d['ms_per_step'])" >> gpurun_out/r2_ab.txt
TG_GEMM_X3=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('x3=1', d['ms_per_step'])" >> gpurun_out/r2_ab.txt
done
cat gpurun_out/r2_t4.log gpurun_out/r2_ab.txt; grep "worst\|assertion" gpurun_out/r2_probe_x3.txt gpurun_out/r2_probe_f32.txt
