export TMPDIR=/tmp
mkdir -p gpurun_out
pass() { tag=$1; sub=$2; shift 2; rm -rf /tmp/pmc_$tag; rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$tag -- python3 tools/h64_pmc.py > /dev/null 2>&1; python3 tools/pmc_summary.py /tmp/pmc_$tag "$sub" 1; }
{
echo "# rocprofv3 --pmc <counters> --output-format csv -- python3 tools/h64_pmc.py   (B=256, T=28, H=64; mover-wave kernels of the final build; first launch skipped)"
echo "## gru_h64_fwd2"; pass h1 gru_h64_fwd2 SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
echo "## gru_h64_bwd2"; pass h2 gru_h64_bwd2 SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
} > gpurun_out/r2_fin3_pmc_gru_h64.txt
cat gpurun_out/r2_fin3_pmc_gru_h64.txt
