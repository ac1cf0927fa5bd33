"""Runs the two mover-wave kernels at the step's biggest shapes: the GRU input projections of both directions as one grouped launch
(gemm_nt_mw_kernel, 2 x [13056 x 900 x 600], weights pre-split) and their weight gradients (gemm_tn_mw_kernel, 2 x [4352 x 900 x 600]
+ bias): target of rocprofv3 --pmc passes (tools/r3_pmc.sh)."""
import importlib, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
M, N, K = 13056, 900, 600
x = torch.randn(M, K, device=dev)
ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(2)]
bs = [torch.randn(N, device=dev) for _ in range(2)]
outs = [torch.empty(M, N, device=dev) for _ in range(2)]
probs = [dict(A=Win.plain(x), W=w, bias=b, out=o, w_planes=ops.split3_planes(w)) for w, b, o in zip(ws, bs, outs)]
assert ops.nt_kernel_plan(probs)[0] == 2, ops.nt_kernel_plan(probs)
for _ in range(6): ops.gemm_nt_group(probs)
Mt = 4352
dys = [torch.randn(Mt, N, device=dev) for _ in range(2)]
xt = torch.randn(Mt, K, device=dev)
dws, dbs = [torch.zeros(N, K, device=dev) for _ in range(2)], [torch.zeros(N, device=dev) for _ in range(2)]
tn = [dict(dY=dy, A=Win.plain(xt), dW=dw, dbias=db) for dy, dw, db in zip(dys, dws, dbs)]
for _ in range(6): ops.gemm_tn_group(tn)
torch.cuda.synchronize()
