#!/bin/bash
# round 5, second half: final measurements on ONE box (run on the GPU box from the repo root): profile + default bench line (with its secondary block) +
# the same-box A/B of this half's schedule changes (fused discriminator head, work on the forward's forked branch, weight-gradient side rows: all off vs
# default) + variants + PMC of the dominant kernel + the GPU suite
set -u
export TMPDIR=/tmp
o=gpurun_out
BENCH_ARGS=--no-secondary bash tools/r5_profile.sh r5_fin2 > $o/r5_fin2_profile.log 2>&1
python3 bench.py > $o/r5_fin2_bench_default.json 2> $o/r5_fin2_bench_default.err
echo "default: $(head -c 200 $o/r5_fin2_bench_default.json)"
for v in "epoch0:--epoch 0 --no-cpu-baseline --no-secondary" "host:--host-input --no-cpu-baseline --no-secondary" "det:--deterministic --no-cpu-baseline --no-secondary" "b256:--batch 256 --no-cpu-baseline --no-secondary" "ddp:--force-ddp --no-cpu-baseline --no-secondary"; do
  tag=${v%%:*}; args=${v#*:}
  python3 bench.py --steps 100 --warmup 20 $args > $o/r5_fin2_bench_$tag.json 2> $o/r5_fin2_bench_$tag.err
  echo "$tag: $(python3 -c "import json;d=json.load(open('$o/r5_fin2_bench_$tag.json'));print(round(d['ms_per_step'],3),'ms',round(d['value']),d['unit'])")"
done
bash tools/ab_env2.sh r5_fin2 3 "TG_D_HEAD_FUSED=0 TG_EARLY_SIDE_WORK=0 TG_TN_SIDE=0" "TG_TN_SIDE=0" "-"
bash tools/r4_pmc.sh > /dev/null 2>&1; cp $o/r4_pmc_gru_fwd_cluster_x3.txt $o/r5_fin2_pmc_gru_fwd_cluster_x3.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=8 > $o/r5_fin2_gpu_tests.txt 2>&1
tail -14 $o/r5_fin2_gpu_tests.txt
