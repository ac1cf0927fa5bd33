"""Runs the big GEMM (GRU input projection shape) in both math modes: target of rocprofv3 --pmc passes."""
import importlib, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
M, N, K = 13056, 900, 600
x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev)
for mode in ("f32", "bf16"):
    ops.set_math_mode(mode)
    for _ in range(6): ops.gemm_nt(Win.plain(x), w, b, out)
torch.cuda.synchronize()
