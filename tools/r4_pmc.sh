#!/bin/bash
# round 4: rocprofv3 --pmc passes of the generator's forward recurrence (one counter group per pass, never combined with tracing), run on the GPU
# box from the repo root:   bash tools/r4_pmc.sh  ->  gpurun_out/r4_pmc_gru_fwd_cluster_x3.txt
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
pass() {   # pass <tag> <script> <kernel substring> <counters...>
  tag=$1; script=$2; sub=$3; shift 3
  rm -rf /tmp/pmc_$tag
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$tag -- python3 $script > /dev/null 2>&1
  python3 tools/pmc_summary.py /tmp/pmc_$tag "$sub" 1
}
{
echo "# rocprofv3 --pmc <counters> --output-format csv -- python3 tools/gru_pmc.py   (separate passes; B=384, H=300, T=34, gates saved for rows [128, 256) only; first launch skipped; round-4 build: same-XCD plain-store hand-off)"
pass g1 tools/gru_pmc.py gru_seq_fwd_cluster FETCH_SIZE
pass g2 tools/gru_pmc.py gru_seq_fwd_cluster WRITE_SIZE
pass g3 tools/gru_pmc.py gru_seq_fwd_cluster SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
} > $out/r4_pmc_gru_fwd_cluster_x3.txt
cat $out/r4_pmc_gru_fwd_cluster_x3.txt
