"""Timing ablation of the generator's cluster-synchronised forward recurrence (lab library, TG_XC_ABL selects a compile-time variant of
gru_seq_fwd_cluster_x3_kernel<2, 2> (round 6: the fp16 x 2 instantiation); ablated launches compute garbage by construction): what is on the step's dependent chain?"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
lab = os.path.join(os.path.dirname(pkg._lib.LIB_PATH), "libtrimodal_hip_lab.so")
assert os.path.exists(lab), "build the lab library first (make lab)"
pkg._lib.LIB_PATH = lab
ops = pkg.ops
dev = torch.device("cuda:0")
T, H, B = 34, 300, 384
g = torch.Generator().manual_seed(1)
gi = (torch.randn(2, B, T, 3 * H, generator=g) * 0.5).to(dev)
w = [(torch.randn(3 * H, H, generator=g) * 0.08).to(dev) for _ in range(2)]
b = [(torch.randn(3 * H, generator=g) * 0.05).to(dev) for _ in range(2)]
y = torch.empty(B, T, 2 * H, device=dev); sv = torch.empty(2, B, T, 4 * H, device=dev)
def timed(iters=30):
    for _ in range(3): ops.gru_forward(gi, w, b, y, sv, save_rows=(128, 128))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gru_forward(gi, w, b, y, sv, save_rows=(128, 128))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
names = {0: "full", 1: "no flag wait", 2: "no fragment loads", 4: "no MFMAs", 8: "no K-slice reduction / gates", 16: "no publishing stores", 32: "no output stores / prefetch",
         64: "no drain before the flag", 80: "no publishing stores, no drain", 3: "no wait, no loads", 7: "no wait / loads / MFMAs", 15: "+ no reduction / gates",
         31: "+ no publishing stores", 128: "no next-step gi prefetch (output stores kept)", 256: "no output stores (prefetch kept)", 127: "everything off: barriers + LDS only", 6: "no loads, no MFMAs", 96: "no output stores, no drain", 81: "no wait, no publish, no drain"}
for rnd in range(2):
    for abl in (0, 1, 2, 4, 8, 16, 32, 128, 256, 64, 80, 96, 3, 6, 7, 15, 31, 81, 127):
        os.environ["TG_XC_ABL"] = str(abl)
        t = timed()
        try:
            ops.check_async_errors()
        except RuntimeError as e:
            print("  (timeout word set:", str(e)[:60], ")")
        print(f"round {rnd} ABL {abl:3d} {names[abl]:40s} {t:7.1f} us  ({t / T * 1000:5.0f} ns per step)")
os.environ["TG_XC_ABL"] = "0"
