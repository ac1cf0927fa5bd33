"""Times the discriminator's H = 64 recurrence kernels (one layer, B = 128 / 256 clips, T = 28), with and without the fused dropout
operands, for both workgroup sizes of the mover-wave kernels (TG_H64_ROWS = 8 | 16 batch rows per workgroup)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
T, H = 28, 64
for B in (128, 256):
    gi = torch.randn(2, B, T, 3 * H, device=dev) * 0.1
    w = [torch.randn(3 * H, H, device=dev) * 0.1 for _ in range(2)]
    wt = [x.t().contiguous() for x in w]
    b = [torch.randn(3 * H, device=dev) * 0.05 for _ in range(2)]
    y = torch.empty(B, T, 2 * H, device=dev); sv = torch.empty(2, B, T, 4 * H, device=dev)
    yd = torch.empty_like(y); mask = (torch.rand(B, T, 2 * H, device=dev) > 0.3).float() / 0.7
    dy = torch.randn(B, T, 2 * H, device=dev)
    dgi, dgh = torch.empty(2, B, T, 3 * H, device=dev), torch.empty(2, B, T, 3 * H, device=dev)
    scratch = torch.empty(4 * B * H, device=dev)
    def timed(fn, iters=100):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters
    for rnd in range(2):
        for rows in ("8", "16"):
            os.environ["TG_H64_ROWS"] = rows
            f0 = timed(lambda: ops.gru_forward(gi, w, b, y, sv))
            f1 = timed(lambda: ops.gru_forward(gi, w, b, y, sv, drop_mask=mask, y_drop=yd))
            b0 = timed(lambda: ops.gru_backward(dy, y, sv, wt, dgi, dgh, scratch))
            b1 = timed(lambda: ops.gru_backward(dy, y, sv, wt, dgi, dgh, scratch, dy_mask=mask))
            print("B=%3d rows/wg %2s  fwd %6.1f us (%.2f us/step)  fwd+dropout %6.1f  bwd %6.1f (%.2f us/step)  bwd+mask %6.1f" % (B, rows, f0, f0 / T, f1, b0, b0 / T, b1))
