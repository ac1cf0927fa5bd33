"""Where a step of the mover-wave H = 64 recurrence kernels (gru_h64_fwd2 / bwd2) spends its time: s_memtime stamps of workgroup (0, 0),
recurrence wave 0 at the phase boundaries (lab library: make -C <package>/csrc lab).  Stamps perturb the step (each drains the scalar-memory
and LDS counters); read the SHARES."""
import ctypes, importlib, os, sys, numpy as np, torch
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
wt = [x.t().contiguous() for x in w]
b = [torch.randn(3 * H, device=dev) * 0.05 for _ in range(2)]
y = torch.empty(B, T, 2 * H, device=dev); sv = torch.empty(2, B, T, 4 * H, device=dev)
yd = torch.empty_like(y); mask = (torch.rand(B, T, 2 * H, device=dev) > 0.3).float() / 0.7
dy = torch.randn(B, T, 2 * H, device=dev)
dgi, dgh = torch.empty(2, B, T, 3 * H, device=dev), torch.empty(2, B, T, 3 * H, device=dev)
scratch = torch.empty(4 * B * H, device=dev)
lib = pkg._lib.load()
raw = ctypes.CDLL(lab)
names = ["operand + fragment reads back, MFMAs issued", "MFMA results back", "gate arithmetic", "split + LDS stores (complete)",
         "record stores to LDS (complete)", "barrier"]
runs = (("forward SAVE + DROP", lambda: ops.gru_forward(gi, w, b, y, sv, drop_mask=mask, y_drop=yd)),
        ("forward SAVE only", lambda: ops.gru_forward(gi, w, b, y, sv)),
        ("backward + mask", lambda: ops.gru_backward(dy, y, sv, wt, dgi, dgh, scratch, dy_mask=mask)))
for label, fn in runs:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = np.zeros((64, 8), dtype=np.uint64)
    assert raw.tg_lab_h64_read_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
    st = out[:T, :7].astype(np.int64)
    d = np.diff(st, axis=1)
    nxt = st[1:, 0] - st[:-1, 6]
    per_step = (st[T - 1, 6] - st[4, 0]) / (T - 5)
    print(f"{label}: {per_step:.0f} cycles per step with stamps")
    for i, n in enumerate(names):
        print(f"   {n:50s} {np.median(d[4:, i]):6.0f} cycles")
    print(f"   {'loop back':50s} {np.median(nxt[4:]):6.0f} cycles")
