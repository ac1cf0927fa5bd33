# scratch: command list of a gpurun call (overwritten per session).  The round-end set:
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/tests.log 2>&1; tail -3 gpurun_out/tests.log
bash tools/r2_profile.sh r2_final > gpurun_out/r2_final_profile.log 2>&1; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r2_final_bench.json
