"""Round 5, lab library: what the H = 64 forward recurrence would cost if its mover waves also computed the NEXT layer's input projection
(TG_H64_ABL=128: the work only -- 72 MFMAs, the operand split and six 16-byte stores per mover wave and step; results are garbage)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
lab = os.path.join(os.path.dirname(pkg._lib.LIB_PATH), "libtrimodal_hip_lab.so")
assert os.path.exists(lab), "build the lab library first (make lab)"
pkg._lib.LIB_PATH = lab
ops = pkg.ops
dev = torch.device("cuda:0")
T, H, B = 28, 64, 256
gi = torch.randn(2, B, T, 3 * H, device=dev) * 0.1
w = [torch.randn(3 * H, H, device=dev) * 0.1 for _ in range(2)]
b = [torch.randn(3 * H, device=dev) * 0.05 for _ in range(2)]
y = torch.empty(B, T, 2 * H, device=dev); sv = torch.empty(2, B, T, 4 * H, device=dev)
yd = torch.empty_like(y); mask = (torch.rand(B, T, 2 * H, device=dev) > 0.3).float() / 0.7
def timed(fn, iters=200):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
names = {0: "full", 128: "movers also run a next-layer projection's work (72 MFMAs + split + 6 stores per wave and step)"}
for rnd in range(3):
    for abl in (0, 128):
        os.environ["TG_H64_ABL"] = str(abl)
        t = timed(lambda: ops.gru_forward(gi, w, b, y, sv, drop_mask=mask, y_drop=yd))
        print(f"round {rnd} ABL {abl:3d} {names[abl]:42s} {t:6.1f} us  ({t / T * 1000:5.0f} ns per step)")
os.environ["TG_H64_ABL"] = "0"
