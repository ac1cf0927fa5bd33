"""Timing ablation of the H = 64 forward recurrence (lab library): how much of a step is the ISSUE of its vector-memory instructions?
mode bit 0 drops the six output stores per step, bit 1 the four operand prefetch loads (results are wrong by construction).  The lab
kernel carries stamp code, so compare the modes with each other, not with the product kernel."""
import ctypes, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
lab = os.path.join(os.path.dirname(pkg._lib.LIB_PATH), "libtrimodal_hip_lab.so")
pkg._lib.LIB_PATH = lab
os.environ["TG_H64_MOVERS"] = "0"      # the stamps / ablation switches live in the single-role kernel (the form these tools studied)
ops = pkg.ops
dev = torch.device("cuda:0")
T, H, B = 28, 64, 256
gi = torch.randn(2, B, T, 3 * H, device=dev) * 0.1
w = [torch.randn(3 * H, H, device=dev) * 0.1 for _ in range(2)]
b = [torch.randn(3 * H, device=dev) * 0.05 for _ in range(2)]
y = torch.empty(B, T, 2 * H, device=dev); sv = torch.empty(2, B, T, 4 * H, device=dev)
yd = torch.empty_like(y); mask = (torch.rand(B, T, 2 * H, device=dev) > 0.3).float() / 0.7
pkg._lib.load()
raw = ctypes.CDLL(lab)
def timed(fn, iters=200):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for mode, name in ((0, "full"), (1, "no output stores"), (2, "no prefetch loads"), (3, "no vector memory at all")):
    torch.cuda.synchronize(); assert raw.tg_lab_h64_set_mode(mode) == 0
    t = timed(lambda: ops.gru_forward(gi, w, b, y, sv, drop_mask=mask, y_drop=yd))
    print(f"mode {mode} {name:28s} {t:6.1f} us  ({t / T:.2f} us per step)")
raw.tg_lab_h64_set_mode(0)
