"""Round 5: the discriminator step's head as three launches (d_head_fwd, gan_d_loss, d_head_bwd) against tg_d_head_step (one launch), B = 128
(256 stacked clips), 300 rounds each -- run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
nb, T, H = 128, 28, 64
y = torch.randn(2 * nb, T, 2 * H, device=dev)
w1, b1, w2, b2 = torch.randn(H, device=dev), torch.randn(1, device=dev), torch.randn(T, device=dev), torch.randn(1, device=dev)
gr = [torch.zeros(H, device=dev), torch.zeros(1, device=dev), torch.zeros(T, device=dev), torch.zeros(1, device=dev)]
out, dl, dy = torch.empty(1, device=dev), torch.empty(2 * nb, device=dev), torch.empty_like(y)


def separate():
    l1, logit, _ = ops.d_head_fwd(y, w1, b1, w2, b2)
    ops.gan_d_loss(logit.view(-1)[:nb], logit.view(-1)[nb:], out, dl[:nb], dl[nb:])
    ops.d_head_bwd(dl, y, l1, w1, w2, dy, gr)


def fused():
    ops.d_head_step(y, w1, b1, w2, b2, nb, 1.0 / nb, 1.0 / nb, out, gr)


def fused_input_gradient_only():
    ops.d_head_step(y, w1, b1, w2, b2, nb, 1.0 / nb, 1.0 / nb, None, None)


which = os.environ.get("PROBE", "separate,fused").split(",")
for name, fn in (("separate", separate), ("fused", fused), ("nograds", fused_input_gradient_only)):
    if name not in which:
        continue
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 300 * 1e3:.1f} us per round (eager launches, launch-bound)")
