#!/bin/bash
# round 5: rocprofv3 --pmc passes of the two mover-wave kernels after the mover rework (one counter group per pass, never combined with tracing):
#   bash tools/r5_pmc_mw.sh  ->  gpurun_out/r5_pmc_gemm_mw.txt     (target: tools/mw_pmc.py, as in round 3: nt 2 x [13056 x 900 x 600], tn 2 x [4352 x 900 x 600] + bias)
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
echo "# rocprofv3 --pmc <counters> --output-format csv -- python3 tools/mw_pmc.py   (nt 2 x [13056 x 900 x 600] weights pre-split; tn 2 x [4352 x 900 x 600] + bias, workspace combine; first launch skipped; round-5 build)"
for k in gemm_nt_mw_kernel gemm_tn_mw_kernel tn_mw_reduce_kernel; do
echo "## $k"
pass m1 tools/mw_pmc.py $k FETCH_SIZE
pass m2 tools/mw_pmc.py $k WRITE_SIZE
pass m3 tools/mw_pmc.py $k SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
pass m4 tools/mw_pmc.py $k SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16
done
} > $out/r5_pmc_gemm_mw.txt 2>&1
cat $out/r5_pmc_gemm_mw.txt
