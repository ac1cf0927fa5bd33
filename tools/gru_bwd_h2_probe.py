"""Backward cluster recurrence, fp16 x 2 against bf16 x 3 (TG_GRU_H2=3 / 1, one process each; the mask's bit 2 = the backward kernel): error of dgi / dgh against an fp64 restatement of the
recurrence on the same taped forward, and the time per launch at the bench's group size (B = 128, T = 34, H = 300; HIP events, 50 launches)."""
import importlib, os, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
T, H = 34, 300
def case(B, seed, dy_scale):
    g = torch.Generator(device="cpu").manual_seed(seed)
    gi = (torch.randn(2, B, T, 3 * H, generator=g) * 0.5).to(dev)
    w = [(torch.randn(3 * H, H, generator=g) * 0.08).to(dev) for _ in range(2)]
    b = [(torch.randn(3 * H, generator=g) * 0.05).to(dev) for _ in range(2)]
    y = torch.empty(B, T, 2 * H, device=dev); sv = torch.empty(2, B, T, 4 * H, device=dev)
    ops.gru_forward(gi, w, b, y, sv)
    # per-row magnitudes spread over six decades: what a mean-reduced loss over clips of different scale hands back
    dy = (torch.randn(B, T, 2 * H, generator=g) * dy_scale * torch.logspace(-5, 1, B).view(B, 1, 1)).to(dev)
    return gi, w, b, y, sv, dy
def ref64(w, y, sv, dy):
    """dgi / dgh of both directions in fp64 from the taped gates (sv = r, z, n, hn per step), as the kernel's cell does."""
    B = y.shape[0]
    out_gi, out_gh = [], []
    for d in range(2):
        W = w[d].double()
        yd = y[..., d * H:(d + 1) * H].double(); s = sv[d].double(); dyd = dy[..., d * H:(d + 1) * H].double()
        dh = torch.zeros(B, H, dtype=torch.float64, device=dev)
        dgi = torch.zeros(B, T, 3 * H, dtype=torch.float64, device=dev); dgh = torch.zeros_like(dgi)
        order = range(T - 1, -1, -1) if d == 0 else range(T)
        for t in order:
            tp = t - 1 if d == 0 else t + 1
            hp = yd[:, tp] if 0 <= tp < T else torch.zeros(B, H, dtype=torch.float64, device=dev)
            r, z, n, hn = s[:, t, :H], s[:, t, H:2 * H], s[:, t, 2 * H:3 * H], s[:, t, 3 * H:]
            dht = dyd[:, t] + dh
            dn = dht * (1 - z) * (1 - n * n)
            dz = dht * (hp - n) * z * (1 - z)
            dr = dn * hn * r * (1 - r)
            dgi[:, t] = torch.cat([dr, dz, dn], 1); dgh[:, t] = torch.cat([dr, dz, dn * r], 1)
            dh = dht * z + dgh[:, t] @ W
        out_gi.append(dgi); out_gh.append(dgh)
    return torch.stack(out_gi), torch.stack(out_gh)
print("TG_GRU_H2 =", os.environ.get("TG_GRU_H2", "3"))
for B, sc in ((128, 1.0), (37, 1e-3), (128, 1e3)):
    gi, w, b, y, sv, dy = case(B, 11 + B, sc)
    wt = [x.t().contiguous() for x in w]
    dgi = torch.empty(2, B, T, 3 * H, device=dev); dgh = torch.empty_like(dgi)
    ops.gru_backward(dy, y, sv, wt, dgi, dgh, torch.zeros(4 * B * H, device=dev))
    torch.cuda.synchronize(); ops.check_async_errors()
    r_gi, r_gh = ref64(w, y, sv, dy)
    # error per batch row relative to that row's own largest gradient (rows differ by six decades)
    rowmax = r_gi.abs().amax(dim=(0, 2, 3)).clamp_min(1e-300).view(1, B, 1, 1)
    e_gi = float(((dgi.double() - r_gi).abs() / rowmax).max()); e_gh = float(((dgh.double() - r_gh).abs() / rowmax).max())
    print(f"B={B} dy scale {sc:g}: max |err| / rowmax  dgi {e_gi:.2e}  dgh {e_gh:.2e}")
gi, w, b, y, sv, dy = case(128, 5, 1.0)
wt = [x.t().contiguous() for x in w]
dgi = torch.empty(2, 128, T, 3 * H, device=dev); dgh = torch.empty_like(dgi); scr = torch.zeros(4 * 128 * H, device=dev)
for _ in range(5): ops.gru_backward(dy, y, sv, wt, dgi, dgh, scr)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.gru_backward(dy, y, sv, wt, dgi, dgh, scr)
e1.record(); e1.synchronize()
print(f"B=128 backward recurrence: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch ({e0.elapsed_time(e1) / 50 / T * 1e3:.2f} us/step)")
