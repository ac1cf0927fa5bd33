"""Round 6: the forward cluster recurrence with fp16 x 2 operands (TG_GRU_H2 bit 1, default on) against bf16 x 3 (TG_GRU_H2=0): time per launch at
B = 384 and error of y / saved gates against an fp64 nn.GRU-style recurrence.  One process per setting (the switch is read once)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
T, H = 34, 300


def ref_dir(gi, w, b, reverse):
    B = gi.shape[0]
    h = torch.zeros(B, H, dtype=torch.float64, device=gi.device)
    ys = [None] * T
    for s in range(T):
        t = T - 1 - s if reverse else s
        gh = h @ w.t() + b
        r = torch.sigmoid(gi[:, t, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        ys[t] = h
    return torch.stack(ys, 1)


def main():
    for B in (128, 384):
        g = torch.Generator().manual_seed(B)
        gi = (torch.randn(2, B, T, 3 * H, generator=g) * 0.5).to(dev)
        w = [(torch.randn(3 * H, H, generator=g) * 0.08 * torch.pow(10.0, torch.randint(-2, 2, (3 * H, 1), generator=g).float())).to(dev) for _ in range(2)]
        b = [(torch.randn(3 * H, generator=g) * 0.05).to(dev) for _ in range(2)]
        y = torch.empty(B, T, 2 * H, device=dev)
        sv = torch.empty(2, B, T, 4 * H, device=dev)
        ops.gru_forward(gi, w, b, y, sv)
        torch.cuda.synchronize()
        ops.check_async_errors()
        ref = torch.cat([ref_dir(gi[d].double(), w[d].double(), b[d].double(), d == 1) for d in range(2)], dim=2)
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ops.gru_forward(gi, w, b, y, sv)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 50)
        ts.sort()
        print(f"TG_GRU_H2={os.environ.get('TG_GRU_H2', '3')}  B={B:4d}: {ts[2]:7.1f} us per launch ({ts[2] / T:5.2f} us per step), y error vs fp64 {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
