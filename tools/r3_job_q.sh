#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
timeout -k 10 200 python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --host-input > $out/r3_q_bench_host.json 2> $out/r3_q_host.err; echo "host rc=$?"; head -c 260 $out/r3_q_bench_host.json; echo
timeout -k 10 200 python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --host-input --no-feed-overlap > $out/r3_q_bench_host_noovl.json 2> $out/r3_q_host2.err; echo "host(no overlap) rc=$?"; head -c 260 $out/r3_q_bench_host_noovl.json; echo
timeout -k 10 200 python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $out/r3_q_bench.json 2> $out/r3_q.err; echo "resident rc=$?"; head -c 260 $out/r3_q_bench.json; echo
timeout -k 10 300 python3 bench.py --mode decode --steps 200 --warmup 20 > $out/r3_q_bench_decode.json 2> $out/r3_q_dec.err; echo "decode rc=$?"; head -c 1400 $out/r3_q_bench_decode.json; echo
timeout -k 10 300 python3 bench.py --mode ae --steps 500 --warmup 50 > $out/r3_q_bench_ae.json 2> $out/r3_q_ae.err; echo "ae rc=$?"; head -c 700 $out/r3_q_bench_ae.json; echo
timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --force-ddp > $out/r3_q_bench_ddp.json 2> $out/r3_q_ddp.err; echo "ddp rc=$?"; head -c 260 $out/r3_q_bench_ddp.json; echo; grep -o '"ddp": {[^}]*}' $out/r3_q_bench_ddp.json
timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --epoch 0 > $out/r3_q_bench_warmup.json 2> $out/r3_q_wu.err; echo "warmup rc=$?"; head -c 260 $out/r3_q_bench_warmup.json; echo
