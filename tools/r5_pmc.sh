#!/bin/bash
# round 5: rocprofv3 --pmc passes of the backward's input-gradient products (one counter group per pass, never combined with tracing), run on the
# GPU box from the repo root:   bash tools/r5_pmc.sh [tag]  ->  gpurun_out/<tag>.txt   (default tag r5_pmc_nt_bwd)
export TMPDIR=/tmp
out=gpurun_out
tag=${1:-r5_pmc_nt_bwd}
kern=${2:-gemm_nt_split_kernel}
mkdir -p $out
pass() {   # pass <tag> <script> <kernel substring> <counters...>
  t=$1; script=$2; sub=$3; shift 3
  rm -rf /tmp/pmc_$t
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$t -- python3 $script > /dev/null 2>&1
  python3 tools/pmc_summary.py /tmp/pmc_$t "$sub" 0
}
{
echo "# rocprofv3 --pmc <counters> --output-format csv -- python3 tools/nt_bwd_pmc.py   (separate passes; 9 launches of (a) gru_dx [4352 x 600 x 1800] K-concatenated, then 9 of (b) tcn_dx [4352 x 300 x 600] + gate epilogue; means are over both shapes unless the kernel names differ)"
python3 tools/nt_bwd_pmc.py --time
for k in $kern; do
echo "## $k"
pass n1 tools/nt_bwd_pmc.py $k FETCH_SIZE
pass n2 tools/nt_bwd_pmc.py $k WRITE_SIZE
pass n3 tools/nt_bwd_pmc.py $k SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
pass n4 tools/nt_bwd_pmc.py $k SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16
pass n5 tools/nt_bwd_pmc.py $k SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
done
} > $out/$tag.txt 2>&1
cat $out/$tag.txt
