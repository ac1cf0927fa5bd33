#!/bin/bash
# rocprofv3 --pmc passes (one counter group per pass, never combined with tracing) for the kernels DESIGN.md quotes; run on the GPU box:
#   bash tools/r2_pmc.sh  ->  gpurun_out/r2_pmc_*.txt
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
echo "# rocprofv3 --pmc <counters> --output-format csv -- python3 tools/gru_pmc.py   (separate passes; B=384, H=300, T=34; first launch skipped)"
pass g1 tools/gru_pmc.py gru_seq_fwd_cluster FETCH_SIZE
pass g2 tools/gru_pmc.py gru_seq_fwd_cluster WRITE_SIZE
pass g3 tools/gru_pmc.py gru_seq_fwd_cluster SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
} > $out/r2_pmc_gru_fwd_cluster_x3.txt
{
echo "# rocprofv3 --pmc <counters> --output-format csv -- python3 tools/h64_pmc.py   (B=256, T=28, H=64; first launch skipped)"
for k in gru_h64_fwd gru_h64_bwd; do
echo "## $k"
pass h1 tools/h64_pmc.py $k SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
pass h2 tools/h64_pmc.py $k SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS
done
} > $out/r2_pmc_gru_h64.txt
{
echo "# rocprofv3 --pmc <counters> --output-format csv -- python3 tools/gemm_pmc.py   (nt M=13056 N=900 K=600; tn M=4352 N=900 K=600)"
for k in gemm_nt_split gemm_tn_split; do
echo "## $k"
pass m1 tools/gemm_pmc.py $k FETCH_SIZE
pass m2 tools/gemm_pmc.py $k WRITE_SIZE
pass m3 tools/gemm_pmc.py $k SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
done
} > $out/r2_pmc_gemm_split.txt
cat $out/r2_pmc_gru_fwd_cluster_x3.txt $out/r2_pmc_gru_h64.txt $out/r2_pmc_gemm_split.txt
