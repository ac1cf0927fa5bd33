"""Weight-gradient GEMM timing vs number of m-splits / tile variant.  Used with two temporary environment knobs (TG_TN_SPLITS,
TG_TN_WIDE) that existed in csrc/gemm.hip while tuning tn_plan(); the knobs are gone, the measured table is quoted there."""
import importlib, os, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
def t(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for (M, N, K) in ((4352, 900, 600), (4352, 900, 300), (4352, 300, 600), (4352, 900, 108), (7168, 192, 128)):
    dy, x = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    for wide in ("0", "1"):
        os.environ["TG_TN_WIDE"] = wide
        res = []
        for sp in ("4", "6", "8", "10", "12", "14", "16", "20", "24", "32"):
            os.environ["TG_TN_SPLITS"] = sp
            us = t(lambda: ops.gemm_tn(dy, Win.plain(x), dW, dbias=db))
            res.append(f"{sp}:{us:6.1f}")
        print(f"tn M={M} N={N} K={K} wide={wide}  " + " ".join(res))
