"""In-situ ablation of the bf16 x 3 NT kernel (128 x 96 tile) on the grouped GRU input projection [13056 x 1800 x 600]: which phase costs what,
and how much of it overlaps.  Needs the lab library (make -C <package>/csrc lab): TG_NT_ABL bit 0 drops the MFMAs, bit 1 the split
arithmetic, bit 2 the LDS fragment reads, bit 3 the global operand loads.  Ablated launches compute garbage by construction."""
import importlib, os, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
lab = os.path.join(os.path.dirname(pkg._lib.LIB_PATH), "libtrimodal_hip_lab.so")
assert os.path.exists(lab), "build the lab library first: make -C gesture-generation-from-trimodal-context_amd/csrc lab"
pkg._lib.LIB_PATH = lab
os.environ["TG_NT_FAST"] = "0"          # the ablated instantiations are of the generic loop
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
def t(fn, iters=1000):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
M, N, K = 13056, 1800, 600
x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.zeros(N, device=dev)
out = torch.empty(M, N, device=dev)
names = {0: "full kernel", 1: "no MFMA", 2: "no split arithmetic", 3: "no MFMA, no split", 4: "no fragment reads", 5: "no MFMA, no fragment reads",
         7: "loads + LDS stores only", 8: "no global loads", 9: "no MFMA, no global loads", 11: "no MFMA, no split, no global loads", 15: "LDS stores + barriers + epilogue only"}
for abl in (0, 1, 2, 3, 4, 5, 7, 8, 9, 11, 15, 0):
    os.environ["TG_NT_ABL"] = str(abl)
    us = t(lambda: ops.gemm_nt(Win.plain(x), w, b, out))
    print(f"ABL {abl:2d}  {names[abl]:40s} {us:7.1f} us", flush=True)
