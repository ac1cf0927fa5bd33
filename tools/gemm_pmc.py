"""Runs the two biggest GEMM shapes of the step on the split-bf16 kernels: forward projection (nt, M = 13056, N = 900, K = 600) and its
weight gradient (tn, M = 4352, N = 900, K = 600): target of rocprofv3 --pmc passes."""
import importlib, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
M, N, K = 13056, 900, 600
x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev)
for _ in range(6): ops.gemm_nt(Win.plain(x), w, b, out)
Mt = 4352
dy, xt = torch.randn(Mt, N, device=dev), torch.randn(Mt, K, device=dev)
dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
for _ in range(6): ops.gemm_tn(dy, Win.plain(xt), dw, dbias=db)
torch.cuda.synchronize()
