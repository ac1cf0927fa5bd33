"""Shape sweep of the cluster recurrences (forward + backward, fp16 x 2) against the per-step launches: hidden sizes other than the model's 300 (member
counts 3 .. 10, ragged last members, waves without a k-step), few rows, short sequences."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
bad = n = 0
for H in (68, 100, 128, 200, 256, 300, 320):
    for B in (1, 17, 64, 130):
        for T in (2, 5, 34):
            if ops.gru_cluster_chunks(B, H) is None or ops.gru_cluster_chunks(B, H, bwd=True) is None:
                continue
            g = torch.Generator().manual_seed(H * 1000 + B * 10 + T)
            w = [(torch.randn(3 * H, H, generator=g) * (1.2 / H ** 0.5)).to(dev) for _ in range(2)]
            b = [(torch.randn(3 * H, generator=g) * 0.05).to(dev) for _ in range(2)]
            wt = [x.t().contiguous() for x in w]
            gi = (torch.randn(2, B, T, 3 * H, generator=g) * 0.5).to(dev)
            dy = (torch.randn(B, T, 2 * H, generator=g) * torch.logspace(-3, 1, B).view(B, 1, 1)).to(dev)
            out = {}
            for cluster in (False, True):
                ops.GRU_CLUSTER = cluster
                y = torch.full((B, T, 2 * H), float("nan"), device=dev); sv = torch.full((2, B, T, 4 * H), float("nan"), device=dev)
                dgi = torch.full((2, B, T, 3 * H), float("nan"), device=dev); dgh = torch.full_like(dgi, float("nan"))
                ops.gru_forward(gi, w, b, y, sv)
                ops.gru_backward(dy, y, sv, wt, dgi, dgh, torch.zeros(4 * B * H, device=dev))
                out[cluster] = (y, sv, dgi, dgh)
            ops.check_async_errors()
            n += 1
            rowmax = out[False][2].abs().amax(dim=(0, 2, 3)).clamp_min(1e-30).view(1, B, 1, 1)
            errs = [float((out[False][0] - out[True][0]).abs().nan_to_num(1e9).max()), float((out[False][1] - out[True][1]).abs().nan_to_num(1e9).max()),
                    float(((out[False][2] - out[True][2]).abs().nan_to_num(1e9) / rowmax).max()), float(((out[False][3] - out[True][3]).abs().nan_to_num(1e9) / rowmax).max())]
            if not (errs[0] < 3e-6 and errs[1] < 2e-5 and errs[2] < 3e-6 and errs[3] < 3e-6):
                bad += 1
                print(f"H={H} B={B} T={T}: y {errs[0]:.2e} gates {errs[1]:.2e} dgi/rowmax {errs[2]:.2e} dgh/rowmax {errs[3]:.2e}", flush=True)
ops.GRU_CLUSTER = True
print(f"{n} shapes, {bad} mismatches (forward outputs and saved gates absolute, gradients relative to the batch row's largest)")
