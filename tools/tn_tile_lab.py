"""Times the weight-gradient products of the training step (grouped as layers.py issues them) for one TG_TN_TILE setting (env; one
process per setting: the choice is read once).  Usage on the GPU box:  TG_TN_TILE=44 python3 tools/tn_tile_lab.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev)


def group(shapes):
    """shapes: list of (M, N, K)"""
    probs, refs = [], []
    for M, N, K in shapes:
        dy, x = rnd(M, N), rnd(M, K)
        dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        probs.append(dict(dY=dy, A=Win.plain(x), dW=dw, dbias=db))
        refs.append((dy, x, dw))
    return probs, refs


def bench(name, shapes, iters=30):
    probs, refs = group(shapes)
    ops.gemm_tn_group(probs)
    torch.cuda.synchronize()
    err = 0.0
    for dy, x, dw in refs:
        ref = dy.double().t() @ x.double()
        err = max(err, float((dw.double() - ref).abs().max() / ref.abs().max()))
    for _ in range(5):
        ops.gemm_tn_group(probs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm_tn_group(probs)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = sum(2.0 * M * N * K for M, N, K in shapes)
    print("%-34s %8.1f us  %6.1f TFLOP/s  err %.1e" % (name, us, fl / us * 1e-6, err))


print("TG_TN_TILE =", os.environ.get("TG_TN_TILE", "(default)"), " TG_GEMM_X3 =", os.environ.get("TG_GEMM_X3", "1"))
bench("gru layer 1-3 (4 x M=4352)", [(4352, 900, 600), (4352, 900, 300)] * 2)
bench("gru layer 0 (K=108)", [(4352, 900, 108), (4352, 900, 300)] * 2)
bench("single 4352x900x600", [(4352, 900, 600)])
bench("single 4352x900x300", [(4352, 900, 300)])
bench("out mlp 4352x150x300", [(4352, 152, 300)])
bench("tcn 4352x300x600 x1", [(4352, 300, 600)])
bench("d gru (4 x M=7168)", [(7168, 192, 128), (7168, 192, 64)] * 2)
bench("audio conv 16 ch (M=139264)", [(139264, 32, 240)], iters=10)
