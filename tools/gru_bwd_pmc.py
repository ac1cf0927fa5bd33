"""Target of rocprofv3 --pmc passes (tools/r6_pmc2.sh): the backward cluster recurrence at B = 128 (fp16 x 2 exchange, magnitude outputs on) as
layers.gru_stack_bwd calls it, and -- with --vec -- the few-row inference recurrence at one sequence."""
import importlib, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
T, H = 34, 300
g = torch.Generator().manual_seed(1)
w = [(torch.randn(3 * H, H, generator=g) * 0.05).to(dev) for _ in range(2)]
b = [(torch.randn(3 * H, generator=g) * 0.05).to(dev) for _ in range(2)]
if "--vec" in sys.argv:
    gi = (torch.randn(2, 1, T, 3 * H, generator=g) * 0.1).to(dev)
    y = torch.empty(1, T, 2 * H, device=dev)
    for _ in range(6): ops.gru_forward(gi, w, b, y, None)
else:
    B = 128
    gi = (torch.randn(2, B, T, 3 * H, generator=g) * 0.1).to(dev)
    y = torch.empty(B, T, 2 * H, device=dev); sv = torch.empty(2, B, T, 4 * H, device=dev)
    ops.gru_forward(gi, w, b, y, sv)
    dy = (torch.randn(B, T, 2 * H, generator=g) * 1e-3).to(dev)
    wt = [x.t().contiguous() for x in w]
    dgi = torch.empty(2, B, T, 3 * H, device=dev); dgh = torch.empty_like(dgi)
    stats = (torch.zeros(2, B, device=dev), torch.zeros(2, 3 * H, device=dev), torch.zeros(2, 3 * H, device=dev))
    for _ in range(6): ops.gru_backward(dy, y, sv, wt, dgi, dgh, torch.zeros(4 * B * H, device=dev), stats=stats)
torch.cuda.synchronize()
