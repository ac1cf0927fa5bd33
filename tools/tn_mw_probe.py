"""Same-process A/B of the big weight-gradient groups: mover-wave kernel (csrc/gemm_tn_mw.hip) against the staged-slab kernel (TG_TN_MW=0 needs
a second process: the switch is read once), sustained launches on random data.  Groups: the four weight gradients of a GRU layer at B = 128
(layers 1-3: K = 600 / 300; layer 0: K = 108 stays on the old kernel) and the text encoder's eight conv gradients."""
import importlib, os, statistics, sys, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
M, H = 4352, 300
dgi = [torch.randn(M, 3 * H, device=dev) * 0.01 for _ in range(2)]
x, hp = torch.randn(M, 2 * H, device=dev), torch.randn(M, H, device=dev)
gru = [dict(dY=dgi[d], A=Win.plain(A), dW=torch.zeros(3 * H, A.shape[1], device=dev), dbias=torch.zeros(3 * H, device=dev)) for d in range(2) for A in (x, hp)]
B, T, C = 128, 34, 300
xs = torch.randn(B, T, C, device=dev)
tcn = [dict(dY=torch.randn(B * T, C, device=dev) * 0.01, A=Win.conv(xs, 2, pad=2 ** (j // 2), dil=2 ** (j // 2), rows_out=T), dW=torch.zeros(C, 2 * C, device=dev),
            dbias=torch.zeros(C, device=dev)) for j in range(8)]
print(f"# tools/tn_mw_probe.py TG_TN_MW={os.environ.get('TG_TN_MW', '1')}: us per grouped launch (median of 7 x 50), fp32-equivalent TFLOP/s")
for name, probs, flops in (("gru layer: 4 weight gradients", gru, 2.0 * M * 900 * (600 + 300) * 2), ("text encoder: 8 conv gradients", tcn, 2.0 * M * 300 * 600 * 8)):
    for _ in range(5): ops.gemm_tn_group(probs)
    us = statistics.median(timed(lambda: ops.gemm_tn_group(probs), 50) for _ in range(7))
    print(f"{name:34s} plan {ops.tn_kernel_plan(probs)}  {us:7.1f} us  {flops / us / 1e6:6.1f} TF", flush=True)
