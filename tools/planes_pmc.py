"""Runs the GRU input projection (both directions grouped: 2 x [13056 x 900 x 600]) on the pre-split kernel (gemm_nt_planes) and, for comparison, on
the split-while-staging kernel (gemm_nt_split): target of rocprofv3 --pmc passes and of a plain timing run (prints us per launch)."""
import importlib, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
M, N, K = 13056, 900, 600
x = torch.randn(M, K, device=dev)
w = [torch.randn(N, K, device=dev) * 0.05 for _ in range(2)]
b = [torch.randn(N, device=dev) for _ in range(2)]
out = torch.empty(2, M, N, device=dev)
a_pl = ops.split3_planes(x)
w_pl = [ops.split3_planes(wi) for wi in w]
def planes(): ops.gemm_nt_planes_group([dict(A=a_pl, Bp=w_pl[i], bias=b[i], out=out[i]) for i in range(2)])
def split(): ops.gemm_nt_group([dict(A=Win.plain(x), W=w[i], bias=b[i], out=out[i]) for i in range(2)])
def t(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
print(f"planes {t(planes):.1f} us   split {t(split):.1f} us   ({2 * 2 * M * N * K / 1e6:.0f} MFLOP per launch)")
