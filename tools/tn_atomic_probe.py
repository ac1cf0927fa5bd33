"""How much of the weight-gradient GEMM is the float-atomic epilogue?  Same kernel with the atomics compiled out (wrong results)."""
import importlib, os, sys, torch
sys.path.insert(0, '/root/repo')
lib_mod = importlib.import_module("gesture-generation-from-trimodal-context_amd._lib")
if len(sys.argv) > 1: lib_mod.LIB_PATH = os.path.abspath(sys.argv[1])
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
def t(fn, iters=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for (M, N, K) in ((4352, 900, 600), (4352, 900, 300), (4352, 300, 600)):
    dy, x = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    dW = torch.zeros(N, K, device=dev)
    res = []
    for cfg in ("0", "1", "2"):
        os.environ["TG_TN_CFG"] = cfg
        dW.zero_(); ops.gemm_tn(dy, Win.plain(x), dW)
        err = float((dW - dy.t() @ x).abs().max() / (dy.t() @ x).abs().max())
        res.append(f"cfg{cfg}: {t(lambda: ops.gemm_tn(dy, Win.plain(x), dW)):.1f} us (err {err:.1e})")
    print(f"tn M={M} N={N} K={K}: " + "  ".join(res))
