"""Runs the two mover-wave kernels at the step's biggest shapes: the GRU input projections of both directions as one grouped launch
(gemm_nt_mw_kernel, 2 x [13056 x 900 x 600], weights pre-split) and their weight gradients (gemm_tn_mw_kernel, 2 x [4352 x 900 x 600]
+ bias): target of rocprofv3 --pmc passes (tools/r3_pmc.sh)."""
import importlib, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
import os
M, N, K = 13056, int(os.environ.get("TG_PMC_N", "900")), 600      # TG_PMC_N=896: rows of 3 584 bytes (64-byte aligned) -- calibrates WRITE_SIZE against N = 900's 3 600-byte rows
x = torch.randn(M, K, device=dev)
ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(2)]
bs = [torch.randn(N, device=dev) for _ in range(2)]
outs = [torch.empty(M, N, device=dev) for _ in range(2)]
# round 6: the planes of the active operand format (fp16 x 2 unless TG_GEMM_H2=0); the activation rows bounded like a GRU layer's output (no pre-pass)
probs = [dict(A=Win.plain(x), W=w, bias=b, out=o, w_planes=ops.split_planes(w)) for w, b, o in zip(ws, bs, outs)]
if ops.gemm_h2():
    sc = ops.h2_row_scales(Win.plain(x))
    probs = [dict(p, a_row_scale=sc) for p in probs]
assert ops.nt_kernel_plan(probs)[0] == 2, ops.nt_kernel_plan(probs)
for _ in range(6): ops.gemm_nt_group(probs)
Mt = 4352
dys = [torch.randn(Mt, N, device=dev) for _ in range(2)]
xt = torch.randn(Mt, K, device=dev)
dws, dbs = [torch.zeros(N, K, device=dev) for _ in range(2)], [torch.zeros(N, device=dev) for _ in range(2)]
tn = [dict(dY=dy, A=Win.plain(xt), dW=dw, dbias=db) for dy, dw, db in zip(dys, dws, dbs)]
if ops.gemm_h2():
    xc = ops.absmax_rows_cols(xt)[1].view(-1)
    tn = [dict(p, y_colmax=ops.absmax_rows_cols(p["dY"])[1].view(-1), a_colmax=xc) for p in tn]
for _ in range(6): ops.gemm_tn_group(tn)
torch.cuda.synchronize()
