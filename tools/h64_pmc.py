"""Runs the discriminator's H = 64 recurrence kernels (one layer, B = 256 stacked real + fake clips, T = 28) a few times: target of rocprofv3 --pmc passes."""
import importlib, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
T, H, B = 28, 64, 256
gi = torch.randn(2, B, T, 3 * H, device=dev) * 0.1
w = [torch.randn(3 * H, H, device=dev) * 0.1 for _ in range(2)]
wt = [x.t().contiguous() for x in w]
b = [torch.randn(3 * H, device=dev) * 0.05 for _ in range(2)]
y = torch.empty(B, T, 2 * H, device=dev); sv = torch.empty(2, B, T, 4 * H, device=dev)
dy = torch.randn(B, T, 2 * H, device=dev)
dgi, dgh = torch.empty(2, B, T, 3 * H, device=dev), torch.empty(2, B, T, 3 * H, device=dev)
for _ in range(5):
    ops.gru_forward(gi, w, b, y, sv)
    ops.gru_backward(dy, y, sv, wt, dgi, dgh, torch.empty(4 * B * H, device=dev))
torch.cuda.synchronize()
