import importlib, os, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
def t(fn, iters=50):
    """kernel time per call from a hipGraph replay of `iters` back-to-back launches (eager launches are host-bound at ~10 us)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for (M, N, K) in ((7168, 192, 128), (7168, 128, 192), (3584, 192, 128), (3584, 128, 192), (4608, 64, 96), (28032, 64, 192), (13056, 150, 300), (4352, 300, 152), (4352, 108, 900), (13056, 64, 300)):
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
    ref = x.double() @ w.double().t() + b.double()
    out = torch.empty(M, N, device=dev)
    res = []
    for knob in (None, "1"):
        if knob: os.environ["TG_NT_BK64"] = knob   # knob of a temporary build, see csrc/gemm.hip
        else: os.environ.pop("TG_NT_BK64", None)
        ops.gemm_nt(Win.plain(x), w, b, out)
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        res.append(f"{'bk64' if knob else 'bk16'}: {t(lambda: ops.gemm_nt(Win.plain(x), w, b, out)):6.1f} us (err {err:.0e})")
    print(f"nt M={M:6d} N={N:4d} K={K:4d}  " + "   ".join(res))
