#!/bin/bash
# round 6, final measurements on ONE box (run on the GPU box from the repo root): profile + default bench line (with its secondary block) + the
# same-box A/B of the round's arithmetic change (TG_GEMM_H2=0 TG_GRU_H2=0 TG_H64_H2=0 = every product back on bf16 x 3) + variants + the GPU suite
set -u
export TMPDIR=/tmp
o=gpurun_out
bash tools/r6_profile.sh r6_fin > $o/r6_fin_profile.log 2>&1
python3 bench.py > $o/r6_fin_bench_default.json 2> $o/r6_fin_bench_default.err
echo "default: $(head -c 240 $o/r6_fin_bench_default.json)"
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value']), round(d['roofline']['frac'],3))"; }
: > $o/r6_fin_ab_h2.txt
for r in 1 2 3; do
  a=$(TG_GEMM_H2=0 TG_GRU_H2=0 TG_H64_H2=0 python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | ms)
  b=$(python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | ms)
  echo "round $r  bf16 x 3 everywhere (ms, clips/s, roofline.frac): $a   |   default, fp16 x 2: $b" | tee -a $o/r6_fin_ab_h2.txt
done
for v in "epoch0:--epoch 0 --no-cpu-baseline --no-secondary" "host:--host-input --no-cpu-baseline --no-secondary" "det:--deterministic --no-cpu-baseline --no-secondary" "b256:--batch 256 --no-cpu-baseline --no-secondary" "ddp:--force-ddp --no-cpu-baseline --no-secondary"; do
  tag=${v%%:*}; args=${v#*:}
  python3 bench.py --steps 100 --warmup 20 $args > $o/r6_fin_bench_$tag.json 2> $o/r6_fin_bench_$tag.err
  echo "$tag: $(python3 -c "import json;d=json.loads(open('$o/r6_fin_bench_$tag.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3),'ms',round(d['value']),d['unit'])")" | tee -a $o/r6_fin_variants.txt
done
timeout -k 10 900 python3 -m pytest tests -m gpu -q --durations=8 > $o/r6_fin_gpu_tests.txt 2>&1
tail -14 $o/r6_fin_gpu_tests.txt
