"""Where a step of the H = 64 forward recurrence spends its time: s_memtime stamps of workgroup (0, 0), wave 0 at the phase boundaries
(lab library: make -C <package>/csrc lab).  Stamps perturb the step (each drains the scalar-memory and LDS counters); read the SHARES."""
import ctypes, importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
lab = os.path.join(os.path.dirname(pkg._lib.LIB_PATH), "libtrimodal_hip_lab.so")
assert os.path.exists(lab), "build the lab library first (make lab)"
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
lib = pkg._lib.load()
raw = ctypes.CDLL(lab)
names = ["LDS fragment reads + MFMA issue", "issue of the previous step's stores", "MFMA results back", "gate arithmetic",
         "split h_t + LDS stores (complete)", "issue of the next prefetch", "barrier"]
for label, kw in (("SAVE + DROP (training forward)", dict(drop_mask=mask, y_drop=yd)), ("SAVE only", {})):
    for _ in range(3):
        ops.gru_forward(gi, w, b, y, sv, **kw)
    torch.cuda.synchronize()
    out = np.zeros((64, 8), dtype=np.uint64)
    assert raw.tg_lab_h64_read_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
    st = out[:T].astype(np.int64)
    d = np.diff(st, axis=1)                                  # [T][7] phase lengths in s_memtime ticks (100 MHz)
    nxt = st[1:, 0] - st[:-1, 7]                             # barrier exit -> next step's first stamp
    per_step = (st[T - 1, 7] - st[4, 0]) / (T - 5)
    print(f"{label}: {per_step:.0f} cycles per step with stamps (s_memtime counts shader cycles)")
    for i, n in enumerate(names):
        print(f"   {n:45s} {np.median(d[4:, i]):6.0f} cycles")
    print(f"   {'loop back':45s} {np.median(nxt[4:]):6.0f} cycles")
