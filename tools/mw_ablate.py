"""In-situ ablation of the mover-wave NT kernel (128 x 192 tile) on the grouped GRU input projection 2 x [13056 x 900 x 600]: which role costs
what, and how much of it overlaps.  Needs the lab library (make -C <package>/csrc lab).  TG_MW_ABL bit 0 drops the MFMAs, bit 1 the movers'
split arithmetic + LDS stores, bit 2 the global operand loads, bit 3 the matrix waves' fragment reads, bit 4 the epilogue's global traffic.
Ablated launches compute garbage by construction."""
import importlib, os, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
lab = os.path.join(os.path.dirname(pkg._lib.LIB_PATH), "libtrimodal_hip_lab.so")
assert os.path.exists(lab), "build the lab library first: make -C gesture-generation-from-trimodal-context_amd/csrc lab"
pkg._lib.LIB_PATH = lab
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
def t(fn, iters=500):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
M, N, K = 13056, 900, 600
x = torch.randn(M, K, device=dev)
ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(2)]
outs = [torch.empty(M, N, device=dev) for _ in range(2)]
H2 = os.environ.get("TG_ABL_H2", "0") == "1"          # round 6: the fp16 x 2 instantiation (three MFMAs per product, two planes)
A = Win.plain(x)
if H2:
    sc = ops.h2_row_scales(A)
    probs = [dict(A=A, W=w, bias=None, out=o, w_planes=ops.split2h_planes(w), a_row_scale=sc) for w, o in zip(ws, outs)]
else:
    probs = [dict(A=A, W=w, bias=None, out=o, w_planes=ops.split3_planes(w)) for w, o in zip(ws, outs)]
print("# operands:", "fp16 x 2" if H2 else "bf16 x 3")
assert ops.nt_kernel_plan(probs) == (2, 128, 192)
names = {0: "full kernel", 1: "no MFMA", 2: "movers: no split, no LDS stores", 6: "movers: weight DMAs only", 8: "no weight DMAs",
         14: "movers idle", 15: "barriers + fragment reads + epilogue", 16: "no epilogue traffic", 32: "weight DMAs awaited one step later",
         33: "no MFMA, weight DMAs awaited one step later"}
for abl in (0, 1, 2, 6, 8, 14, 15, 16, 32, 33, 0):
    os.environ["TG_MW_ABL"] = str(abl)
    us = t(lambda: ops.gemm_nt_group(probs))
    print(f"ABL {abl:2d}  {names[abl]:50s} {us:7.1f} us", flush=True)
