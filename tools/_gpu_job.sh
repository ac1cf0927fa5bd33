export TMPDIR=/tmp
mkdir -p gpurun_out
export TG_NT_OCC=1
python3 tools/planes_pmc.py
TG_NP_OCC=2 python3 tools/planes_pmc.py
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "planes" 2>&1 | tail -2
