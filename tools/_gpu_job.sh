mkdir -p gpurun_out
python3 bench.py > gpurun_out/r2_w_bench_full.json 2> gpurun_out/r2_w_bench_full.err; tail -c 600 gpurun_out/r2_w_bench_full.json; echo
python3 bench.py --mode decode --no-cpu-baseline > gpurun_out/r2_w_bench_decode.json 2>> gpurun_out/r2_w_bench_full.err; cut -c1-400 gpurun_out/r2_w_bench_decode.json; echo
python3 bench.py --mode ae --no-cpu-baseline > gpurun_out/r2_w_bench_ae.json 2>> gpurun_out/r2_w_bench_full.err; cut -c1-400 gpurun_out/r2_w_bench_ae.json; echo
python3 bench.py --epoch 0 --no-cpu-baseline > gpurun_out/r2_w_bench_warmup.json 2>> gpurun_out/r2_w_bench_full.err; cut -c1-300 gpurun_out/r2_w_bench_warmup.json; echo
python3 bench.py --host-input --no-cpu-baseline > gpurun_out/r2_w_bench_host.json 2>> gpurun_out/r2_w_bench_full.err; cut -c1-300 gpurun_out/r2_w_bench_host.json; echo
python3 bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r2_w_bench_bf16.json 2>> gpurun_out/r2_w_bench_full.err; cut -c1-300 gpurun_out/r2_w_bench_bf16.json; echo
bash tools/r2_profile.sh r2_w > gpurun_out/r2_w_profile.log 2>&1; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r2_w_bench.json; head -3 gpurun_out/r2_w_timeline.txt
