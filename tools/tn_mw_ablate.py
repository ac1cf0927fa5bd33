"""In-situ ablation of the mover-wave TN (weight-gradient) kernel on the four weight gradients of a GRU layer at B = 128 (4 352 rows; N = 900;
K = 600 / 300, bias columns riding along): which role costs what, and how much of it overlaps.  Needs the lab library
(make -C <package>/csrc lab).  TG_TNMW_ABL bit 0 drops the MFMAs, bit 1 the movers' split arithmetic + LDS stores, bit 2 the movers' global
loads, bit 3 the matrix waves' LDS fragment reads, bit 4 the epilogue's atomics.  Ablated launches compute garbage by construction."""
import importlib, os, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
lab = os.path.join(os.path.dirname(pkg._lib.LIB_PATH), "libtrimodal_hip_lab.so")
assert os.path.exists(lab), "build the lab library first: make -C gesture-generation-from-trimodal-context_amd/csrc lab"
pkg._lib.LIB_PATH = lab
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
def t(fn, iters=300):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
M, H = 4352, 300
dgi = [torch.randn(M, 3 * H, device=dev) * 0.01 for _ in range(2)]
x, hp = torch.randn(M, 2 * H, device=dev), torch.randn(M, H, device=dev)
gru = [dict(dY=dgi[d], A=Win.plain(A), dW=torch.zeros(3 * H, A.shape[1], device=dev), dbias=torch.zeros(3 * H, device=dev)) for d in range(2) for A in (x, hp)]
assert ops.tn_kernel_plan(gru) == 2, ops.tn_kernel_plan(gru)
names = {0: "full kernel", 1: "no MFMA", 2: "movers: no split, no LDS stores", 4: "movers: no global loads", 6: "movers idle (barriers only)",
         8: "matrix waves: no fragment reads", 9: "matrix waves: barriers only", 14: "no movers, no fragment reads", 15: "barriers + epilogue only", 16: "no epilogue atomics"}
for abl in (0, 1, 2, 4, 6, 8, 9, 14, 15, 16, 0):
    os.environ["TG_TNMW_ABL"] = str(abl)
    us = t(lambda: ops.gemm_tn_group(gru))
    print(f"ABL {abl:2d}  {names[abl]:45s} {us:7.1f} us", flush=True)
