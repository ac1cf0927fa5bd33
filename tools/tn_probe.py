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
for (M, N, K) in ((4352, 900, 600), (4352, 900, 300), (4352, 300, 600), (27776, 64, 480), (4352, 32, 960), (7168, 192, 128), (7168, 192, 64), (4352, 150, 300), (168064, 32, 240)):
    dy, x = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    res = []
    for two_pass in (False, True):
        ops.TN_TWO_PASS_ROWS = 1 if two_pass else 1 << 30
        for sp in ("", "4", "8", "16", "32", "64"):
            if sp: os.environ["TG_TN_SPLITS"] = sp
            else: os.environ.pop("TG_TN_SPLITS", None)
            us = t(lambda: ops.gemm_tn(dy, Win.plain(x), dW, dbias=db))
            res.append(f"{'2p' if two_pass else 'at'}{sp or 'def'}:{us:6.1f}")
    print(f"tn M={M} N={N} K={K}  " + " ".join(res))
