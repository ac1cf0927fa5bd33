"""Round 6: does the stacked forward's GRU input projection (mover-wave kernel, fp16 x 2, 2 x [13056 x 900 x K]) write faster into rows of 912 floats
(3 648 bytes = 57 whole 64-byte lines) than into its natural 900 (3 600 bytes: three of four row starts are not line-aligned)?  PMC had counted 178 MB
of write requests for 94 MB of output at 900 and 122.6 MB at 896 (profiles/r6_pmc_gemm_mw.txt)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
M, N = 13056, 900
def timed(fn, n=60, rounds=5):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]
for K in (600, 108):
    x = torch.randn(M, K, device=dev)
    ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(2)]
    bs = [torch.randn(N, device=dev) for _ in range(2)]
    sc = ops.h2_row_scales(Win.plain(x))
    res = {}
    for ld in (900, 912, 928):
        outs = [torch.empty(M, ld, device=dev) for _ in range(2)]
        probs = [dict(A=Win.plain(x), W=w, bias=b, out=o[:, :N], c_row_stride=ld, c_batch_stride=M * ld, c_rows_out=M, w_planes=ops.split_planes(w), a_row_scale=sc)
                 for w, b, o in zip(ws, bs, outs)]
        plan = ops.nt_kernel_plan(probs)
        ops.gemm_nt_group(probs); torch.cuda.synchronize()
        res[ld] = (timed(lambda: ops.gemm_nt_group(probs)), plan)
    print(f"K={K}: " + "   ".join(f"row stride {ld}: {t:6.1f} us (plan {p})" for ld, (t, p) in res.items()), flush=True)
