"""Runs the dominant kernel (generator GRU forward step, B = 384, H = 300) a few times: target of the rocprofv3 --pmc passes."""
import importlib, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
T, H, B = 34, 300, 384
gi = torch.randn(2, B, T, 3 * H, device=dev) * 0.1
w = [torch.randn(3 * H, H, device=dev) * 0.05 for _ in range(2)]
b = [torch.randn(3 * H, device=dev) * 0.05 for _ in range(2)]
y = torch.empty(B, T, 2 * H, device=dev); sv = torch.empty(2, B, T, 4 * H, device=dev)
for _ in range(5): ops.gru_forward(gi, w, b, y, sv, save_rows=(B // 3, B // 3))     # as the trainer calls it: gates saved for the differentiated call only
torch.cuda.synchronize()
