"""Soak of the few-row inference recurrence (csrc/gru_vec.hip): thousands of launches on ONE workspace with random row counts (1 .. 4) and lengths
(1 .. 40), launched back to back (the two exchange buffers alternate, slots rotate, T == 1 launches in between leave the launch counter alone),
every result compared with the cluster kernel's for the same call.  A stale, missed or torn hand-off word shows as an O(1e-2) difference; the
sticky timeout word is checked at the end.
    python tools/gru_vec_soak.py [launches = 3000]"""
import importlib, os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
H = 300
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rnd = random.Random(7)
g = torch.Generator().manual_seed(7)
w = [(torch.randn(3 * H, H, generator=g) * 0.08).to(dev) for _ in range(2)]
b = [(torch.randn(3 * H, generator=g) * 0.05).to(dev) for _ in range(2)]
pool = (torch.randn(2, 4, 40, 3 * H, generator=g) * 0.7).to(dev)
bad, worst, pending = 0, 0.0, []
for i in range(n):
    B, T = rnd.randint(1, 4), rnd.choice([1, 2, 3, 5, 17, 34, 34, 34, 40])
    gi = (pool[:, :B, :T] * (1.0 + 0.001 * (i % 7))).contiguous()
    y = torch.full((B, T, 2 * H), float("nan"), device=dev)
    ops.GRU_VEC = True
    ops.gru_forward(gi, w, b, y, None)
    pending.append((gi, y, B, T))
    if len(pending) == 50 or i == n - 1:          # 50 launches back to back, then the comparisons
        ops.GRU_VEC = False
        for gi_, y_, B_, T_ in pending:
            yc = ops.gru_forward(gi_, w, b, torch.empty_like(y_), None)
            d = float((y_ - yc).abs().nan_to_num(1e9).max())
            worst = max(worst, d)
            if not d < 3e-6:
                bad += 1
                print(f"launch ~{i}: B={B_} T={T_} max |diff| {d:.3e}", flush=True)
        pending = []
ops.check_async_errors()
print(f"{n} launches of tg_gru_forward_vec on one workspace (rows 1..4, T 1..40): {bad} mismatches against the cluster kernel, worst |diff| {worst:.2e}, no timeout")
