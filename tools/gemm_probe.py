"""Timing probe of the GEMM kernels at the shapes of the training step."""
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
for (M, N, K) in ((13056, 900, 600), (13056, 900, 108), (13056, 300, 600), (4352, 600, 900), (4352, 300, 600), (4352, 108, 900), (13056, 150, 300)):
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    us = t(lambda: ops.gemm_nt(Win.plain(x), w, b, out))
    print(f"nt M={M:6d} N={N:4d} K={K:4d}: {us:8.1f} us  {2*M*N*K/us/1e6:6.1f} TFLOP/s")
for (M, N, K) in ((4352, 900, 600), (4352, 900, 300), (4352, 300, 600), (4352, 900, 108)):
    dy, x = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    us = t(lambda: ops.gemm_tn(dy, Win.plain(x), dW, dbias=db))
    print(f"tn M={M:6d} N={N:4d} K={K:4d}: {us:8.1f} us  {2*M*N*K/us/1e6:6.1f} TFLOP/s")
