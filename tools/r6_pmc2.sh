#!/bin/bash
# round 6, second half: rocprofv3 --pmc passes (one counter group per pass, never combined with tracing) of the two kernels that are new in it:
#   bash tools/r6_pmc2.sh -> gpurun_out/r6_pmc_gru_bwd.txt (backward cluster recurrence, fp16 x 2, B = 128) and gpurun_out/r6_pmc_gru_vec.txt (few-row kernel, one sequence)
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
pass() {   # pass <tag> <script + args> <kernel substring> <counters...>
  tag=$1; script=$2; sub=$3; shift 3
  rm -rf /tmp/pmc_$tag
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$tag -- python3 $script > /dev/null 2>&1
  python3 tools/pmc_summary.py /tmp/pmc_$tag "$sub" 1
}
{
echo "# rocprofv3 --pmc <counters> --output-format csv -- python3 tools/gru_bwd_pmc.py   (gru_seq_bwd_cluster_x3_kernel<2>: fp16 x 2 exchange, B = 128, H = 300, T = 34, magnitude outputs on; first launch skipped)"
pass b1 tools/gru_bwd_pmc.py gru_seq_bwd_cluster FETCH_SIZE
pass b2 tools/gru_bwd_pmc.py gru_seq_bwd_cluster WRITE_SIZE
pass b3 tools/gru_bwd_pmc.py gru_seq_bwd_cluster SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
pass b4 tools/gru_bwd_pmc.py gru_seq_bwd_cluster SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VMEM
} > $out/r6_pmc_gru_bwd.txt 2>&1
cat $out/r6_pmc_gru_bwd.txt
{
echo "# rocprofv3 --pmc <counters> --output-format csv -- python3 tools/gru_bwd_pmc.py --vec   (gru_seq_fwd_vec_kernel<1>: one sequence, H = 300, T = 34; first launch skipped)"
pass v1 "tools/gru_bwd_pmc.py --vec" gru_seq_fwd_vec FETCH_SIZE
pass v2 "tools/gru_bwd_pmc.py --vec" gru_seq_fwd_vec WRITE_SIZE
pass v3 "tools/gru_bwd_pmc.py --vec" gru_seq_fwd_vec SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU
pass v4 "tools/gru_bwd_pmc.py --vec" gru_seq_fwd_vec SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU
} > $out/r6_pmc_gru_vec.txt 2>&1
cat $out/r6_pmc_gru_vec.txt
