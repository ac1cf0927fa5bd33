"""Timing + error probe of the GEMM kernels at the shapes of the training step, fp32 tier vs bf16-operand tier."""
import importlib, sys, torch
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
def rel(a, b): return float((a.double() - b).abs().max() / b.abs().max())
for (M, N, K) in ((13056, 900, 600), (13056, 900, 108), (13056, 300, 600), (4352, 600, 900), (4352, 300, 600), (4352, 108, 900), (13056, 150, 300)):
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
    ref = x.double() @ w.double().t() + b.double()
    out = torch.empty(M, N, device=dev)
    res = []
    for mode in ("f32", "bf16"):
        ops.set_math_mode(mode)
        us = t(lambda: ops.gemm_nt(Win.plain(x), w, b, out))
        res.append(f"{mode}: {us:7.1f} us {2*M*N*K/us/1e6:6.1f} TF err {rel(out, ref):.1e}")
    print(f"nt M={M:6d} N={N:4d} K={K:4d}  " + "   ".join(res))
for (M, N, K) in ((4352, 900, 600), (4352, 900, 300), (4352, 300, 600), (4352, 900, 108), (13056, 32, 480)):
    dy, x = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    ref = dy.double().t() @ x.double(); refb = dy.double().sum(0)
    res = []
    for mode in ("f32", "bf16"):
        ops.set_math_mode(mode)
        dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        ops.gemm_tn(dy, Win.plain(x), dW, dbias=db)
        e, eb = rel(dW, ref), rel(db, refb)
        us = t(lambda: ops.gemm_tn(dy, Win.plain(x), dW, dbias=db))
        res.append(f"{mode}: {us:7.1f} us {2*M*N*K/us/1e6:6.1f} TF err {e:.1e} bias {eb:.1e}")
    print(f"tn M={M:6d} N={N:4d} K={K:4d}  " + "   ".join(res))
ops.set_math_mode("f32")
