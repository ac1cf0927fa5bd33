"""Probe of the H=300 GRU step kernel: time per launch vs batch size (is it bound by per-launch weight traffic?)."""
import importlib, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
T, H = 34, 300
def run(B, iters=10):
    gi = torch.randn(2, B, T, 3 * H, device=dev) * 0.1
    w = [torch.randn(3 * H, H, device=dev) * 0.05 for _ in range(2)]
    b = [torch.randn(3 * H, device=dev) * 0.05 for _ in range(2)]
    y = torch.empty(B, T, 2 * H, device=dev); sv = torch.empty(2, B, T, 4 * H, device=dev)
    for _ in range(3): ops.gru_forward(gi, w, b, y, sv)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gru_forward(gi, w, b, y, sv)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters / T
for B in (32, 64, 128, 256, 384, 768):
    print(f"B={B:4d}  {run(B):7.2f} us per step launch")
