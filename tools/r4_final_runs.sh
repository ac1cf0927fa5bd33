set -u
export TMPDIR=/tmp
o=gpurun_out
bash tools/r4_profile.sh r4_fin > $o/r4_fin_profile.log 2>&1
python3 bench.py > $o/r4_fin_bench_default.json 2> $o/r4_fin_bench_default.err
for v in "decode:--mode decode" "decode_b1:--mode decode --batch 1" "ae:--mode ae" "bf16:--dtype bf16 --no-cpu-baseline" "epoch0:--epoch 0 --no-cpu-baseline" "host:--host-input --no-cpu-baseline" "hostrec:--host-records --no-cpu-baseline" "ddp:--force-ddp --no-cpu-baseline" "det:--deterministic --no-cpu-baseline"; do
  tag=${v%%:*}; args=${v#*:}
  python3 bench.py $args > $o/r4_fin_bench_$tag.json 2> $o/r4_fin_bench_$tag.err
  echo "$tag: $(head -c 260 $o/r4_fin_bench_$tag.json)"
done
bash tools/r4_pmc.sh > /dev/null 2>&1
cat $o/r4_pmc_gru_fwd_cluster_x3.txt
