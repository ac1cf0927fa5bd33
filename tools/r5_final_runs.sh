#!/bin/bash
# round 5, final measurements on ONE box (run on the GPU box from the repo root): profile + default bench line (with its secondary block) + the
# same-box A/B against the round-4 library + variants + PMC + the GPU suite
set -u
export TMPDIR=/tmp
o=gpurun_out
bash tools/r5_profile.sh r5_fin > $o/r5_fin_profile.log 2>&1
python3 bench.py > $o/r5_fin_bench_default.json 2> $o/r5_fin_bench_default.err
echo "default: $(head -c 200 $o/r5_fin_bench_default.json)"
for v in "epoch0:--epoch 0 --no-cpu-baseline --no-secondary" "host:--host-input --no-cpu-baseline" "hostrec:--host-records --no-cpu-baseline" "det:--deterministic --no-cpu-baseline" "b256:--batch 256 --no-cpu-baseline --no-secondary" "ddp:--force-ddp --no-cpu-baseline"; do
  tag=${v%%:*}; args=${v#*:}
  python3 bench.py --steps 100 --warmup 20 $args > $o/r5_fin_bench_$tag.json 2> $o/r5_fin_bench_$tag.err
  echo "$tag: $(python3 -c "import json;d=json.load(open('$o/r5_fin_bench_$tag.json'));print(round(d['ms_per_step'],3),'ms',round(d['value']),d['unit'])")"
done
[ -f gesture-generation-from-trimodal-context_amd/libtrimodal_hip_base.so ] && bash tools/ab_bench.sh r5_fin 3
bash tools/r5_pmc_mw.sh > /dev/null 2>&1
bash tools/r4_pmc.sh > /dev/null 2>&1; cp $o/r4_pmc_gru_fwd_cluster_x3.txt $o/r5_pmc_gru_fwd_cluster_x3.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=12 > $o/r5_fin_gpu_tests.txt 2>&1
tail -18 $o/r5_fin_gpu_tests.txt
