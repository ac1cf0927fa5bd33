#!/bin/bash
# round 6: rocprofv3 --pmc passes (one counter group per pass, never combined with tracing) of the kernels whose arithmetic changed to fp16 x 2:
#   bash tools/r6_pmc.sh  ->  gpurun_out/r6_pmc_gru_fwd.txt (tools/gru_pmc.py: the forward cluster recurrence at B = 384, as the trainer calls it)
#                             gpurun_out/r6_pmc_gemm_mw.txt (tools/mw_pmc.py: nt 2 x [13056 x 900 x 600], tn 2 x [4352 x 900 x 600] + bias)
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
echo "# rocprofv3 --pmc <counters> --output-format csv -- python3 tools/gru_pmc.py   (gru_seq_fwd_cluster_x3_kernel<2, 2>: fp16 x 2, B = 384, H = 300, T = 34, gates saved for 128 rows; first launch skipped; round-6 build)"
pass g1 tools/gru_pmc.py gru_seq_fwd_cluster FETCH_SIZE
pass g2 tools/gru_pmc.py gru_seq_fwd_cluster WRITE_SIZE
pass g3 tools/gru_pmc.py gru_seq_fwd_cluster SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
pass g4 tools/gru_pmc.py gru_seq_fwd_cluster SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VMEM
} > $out/r6_pmc_gru_fwd.txt 2>&1
cat $out/r6_pmc_gru_fwd.txt
{
echo "# rocprofv3 --pmc <counters> --output-format csv -- python3 tools/mw_pmc.py   (fp16 x 2: nt 2 x [13056 x 900 x 600] weights pre-split, row scales supplied; tn 2 x [4352 x 900 x 600] + bias, column magnitudes supplied, workspace combine; first launch skipped; round-6 build)"
for k in gemm_nt_mw_kernel gemm_tn_mw_kernel tn_mw_reduce_kernel; do
echo "## $k"
pass m1 tools/mw_pmc.py $k FETCH_SIZE
pass m2 tools/mw_pmc.py $k WRITE_SIZE
pass m3 tools/mw_pmc.py $k SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
pass m4 tools/mw_pmc.py $k SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16
done
echo "## gemm_nt_mw_kernel, N = 896 instead of 900 (output rows of 3 584 bytes: every 64-byte piece of a row inside ONE 64-byte block; 13056 x 896 x 4 x 2 = 93.6 MB written)"
export TG_PMC_N=896
pass m5 tools/mw_pmc.py gemm_nt_mw_kernel WRITE_SIZE
unset TG_PMC_N
} > $out/r6_pmc_gemm_mw.txt 2>&1
cat $out/r6_pmc_gemm_mw.txt
{
echo "# rocprofv3 --pmc <counters> --output-format csv -- python3 tools/nt_bwd_pmc.py   (gru_dx [4352 x 600 x 1800] on gemm_nt_mw_kernel<2, 3, 4, 2, 2>: fp16 x 2, 128 x 96 tiles, 238 of them; round-6 build)"
python3 tools/nt_bwd_pmc.py --time
pass b1 tools/nt_bwd_pmc.py gemm_nt_mw_kernel FETCH_SIZE
pass b2 tools/nt_bwd_pmc.py gemm_nt_mw_kernel WRITE_SIZE
pass b3 tools/nt_bwd_pmc.py gemm_nt_mw_kernel SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
} > $out/r6_pmc_nt_bwd.txt 2>&1
cat $out/r6_pmc_nt_bwd.txt
