"""Is the bf16 x 3 GEMM held down by the chip's clock management?  Same launch on random and on all-zero operands (zeros toggle no
datapath bits: MI355X_MICROARCH.md 'DVFS give-back' item 1), 2000 back-to-back launches each so the clock settles."""
import importlib, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
def t(fn, iters):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
M, N, K = 13056, 1800, 600
for mode in ("f32", "bf16"):
    ops.set_math_mode(mode)
    for kind in ("random", "zeros", "random"):
        x = torch.randn(M, K, device=dev) if kind == "random" else torch.zeros(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05 if kind == "random" else torch.zeros(N, K, device=dev)
        b = torch.zeros(N, device=dev)
        out = torch.empty(M, N, device=dev)
        us = t(lambda: ops.gemm_nt(Win.plain(x), w, b, out), 2000)
        print(f"{mode:5s} {kind:7s} M={M} N={N} K={K}: {us:7.1f} us  {2*M*N*K/us/1e6:6.1f} TFLOP/s", flush=True)
ops.set_math_mode("f32")
