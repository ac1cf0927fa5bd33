"""Timing ablation of the mover-wave H = 64 forward recurrence (lab library, TG_H64_ABL selects a compile-time variant of gru_h64_fwd2_kernel;
ablated launches compute garbage by construction): which part of a step is on the dependent chain?"""
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
names = {0: "full", 1: "movers idle", 2: "no transcendental gate arithmetic", 4: "no MFMAs", 8: "no record stores", 16: "no split / h store",
         32: "no mover lag", 64: "no priority", 96: "no lag, no priority", 3: "movers idle + no gates", 6: "no gates, no MFMAs", 7: "movers idle, no gates, no MFMAs",
         9: "movers idle + no record stores", 15: "idle movers, no gates / MFMAs / record", 22: "no gates / MFMAs / split", 30: "no gates / MFMAs / record / split",
         31: "everything off: reads + barrier"}
for rnd in range(2):
    for abl in (0, 1, 2, 4, 8, 16, 32, 64, 96, 3, 9, 6, 7, 15, 22, 30, 31):
        os.environ["TG_H64_ABL"] = str(abl)
        t = timed(lambda: ops.gru_forward(gi, w, b, y, sv, drop_mask=mask, y_drop=yd))
        print(f"round {rnd} ABL {abl:3d} {names[abl]:42s} {t:6.1f} us  ({t / T * 1000:5.0f} ns per step)")
wt = [x.t().contiguous() for x in w]
dy = torch.randn(B, T, 2 * H, device=dev)
dgi, dgh = torch.empty(2, B, T, 3 * H, device=dev), torch.empty(2, B, T, 3 * H, device=dev)
scratch = torch.empty(4 * B * H, device=dev)
os.environ["TG_H64_ABL"] = "0"
ops.gru_forward(gi, w, b, y, sv, drop_mask=mask, y_drop=yd)
bnames = dict(names); bnames[2] = "no gate arithmetic before the product"
print("backward (bwd2, with mask)")
for rnd in range(2):
    for abl in (0, 1, 2, 4, 8, 16, 32, 3, 9, 6, 7, 15, 30, 31):
        os.environ["TG_H64_ABL"] = str(abl)
        t = timed(lambda: ops.gru_backward(dy, y, sv, wt, dgi, dgh, scratch, dy_mask=mask))
        print(f"round {rnd} ABL {abl:3d} {bnames[abl]:42s} {t:6.1f} us  ({t / T * 1000:5.0f} ns per step)")
os.environ["TG_H64_ABL"] = "0"
