"""Feasibility probe for the backward's GRU input gradient [4352 x 600 x 1800] as a K-split on the mover-wave kernel: q problems of one
grouped launch, each [4352 x 600 x 1800 / q] with pre-split weights, into partial buffers + the fixed-order sum -- against the staged-slab
kernel on the whole product (K-concatenated weight segments), same process, interleaved."""
import importlib, statistics, sys, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
def timed(fn, iters=100):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
M, N, K = 4352, 600, 1800
dgi = torch.randn(2, M, 900, device=dev)
wt = torch.randn(2, N, 900, device=dev) * 0.05
dx = torch.empty(M, N, device=dev)
a_cat = Win(dgi, batches=1, batch_stride=0, row_stride=900, rows_in=2 * M, rows_out=M, cw=900, K=K, dil=M)
base = lambda: ops.gemm_nt(a_cat, wt[0], None, dx, b_seg=(900, N * 900))
res = {}
for q in (2, 3, 4, 6):
    kq = K // q
    xs = [torch.randn(M, kq, device=dev) for _ in range(q)]
    ws = [torch.randn(N, kq, device=dev) * 0.05 for _ in range(q)]
    parts = torch.empty(q, M, N, device=dev)
    probs = [dict(A=Win.plain(x), W=w, bias=None, out=parts[i], w_planes=ops.split3_planes(w)) for i, (x, w) in enumerate(zip(xs, ws))]
    plan = ops.nt_kernel_plan(probs)
    def run(probs=probs, parts=parts):
        ops.gemm_nt_group(probs)
        ops.sum_parts(parts, dx)
    res[q] = (plan, run, lambda probs=probs: ops.gemm_nt_group(probs))
for _ in range(3):
    base()
    for q in res: res[q][1]()
print(f"staged-slab kernel, whole product          {statistics.median(timed(base) for _ in range(5)):7.1f} us")
for q, (plan, run, gem) in res.items():
    print(f"K split {q}: plan {plan}  product {statistics.median(timed(gem) for _ in range(5)):7.1f} us   product + sum_parts {statistics.median(timed(run) for _ in range(5)):7.1f} us")
