"""Training parity at odd batch sizes for several draw seeds, to tell a ReLU-tie flip (one seed off by ~1e-3, the others ~1e-6) from
an arithmetic problem (every seed off).  Run once with TG_GEMM_X3=1 (default) and once with TG_GEMM_X3=0."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
from tests.harness import run_train_parity
dev = torch.device("cuda:0")
print("TG_GEMM_X3 =", os.environ.get("TG_GEMM_X3", "(default 1)"))
for batch in (21, 40):
    for rs in (1017, 2001, 2002, 2003, 2004, 2005):
        try:
            w = run_train_parity(pkg, dev, batch=batch, epochs=(11,), rand_seed=rs, verbose=True, check_step=False)
            print(f"batch {batch} rand_seed {rs}: worst {w:.2e}")
        except AssertionError as e:
            print(f"batch {batch} rand_seed {rs}: assertion {e}")
