"""Per-workgroup timeline of ONE launch of the mover-wave NT kernel (lab library: s_memrealtime stamps, 10 ns ticks) on 2 x [13056 x 900 x 600]:
when workgroups start and end, how long each phase takes, how the rounds on a CU follow each other."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
lab = os.path.join(os.path.dirname(pkg._lib.LIB_PATH), "libtrimodal_hip_lab.so")
pkg._lib.LIB_PATH = lab
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
M, N, K = 13056, 900, 600
x = torch.randn(M, K, device=dev)
ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(2)]
outs = [torch.empty(M, N, device=dev) for _ in range(2)]
probs = [dict(A=Win.plain(x), W=w, bias=None, out=o) for w, o in zip(ws, outs)]
os.environ["TG_MW_ABL"] = sys.argv[1] if len(sys.argv) > 1 else "0"
for _ in range(5):
    ops.gemm_nt_group(probs)
torch.cuda.synchronize()
ops.gemm_nt_group(probs)
torch.cuda.synchronize()
lib = pkg._lib.load()
n = 2048 * 16
buf = (C.c_uint64 * n)()
assert lib.tg_lab_mw_stamps(buf, n) == 0
import numpy as np
st = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 2, 8).astype(np.int64)
live = st[:, 0, 0] > 0
st = st[live]
t0 = st[:, :, 0].min()
tick = 0.01  # us
print(f"# ABL={os.environ['TG_MW_ABL']}  workgroups stamped: {len(st)}")
names = ["entry", "setup done", "first barrier", "loop done", "tile laid out", "end"]
for role, rn in ((0, "matrix wave 0"), (1, "mover wave 4")):
    d = np.diff(st[:, role, :6], axis=1) * tick
    print(f"{rn}: phase durations us (median / p90 / max): " + "  ".join(f"{names[i]}->{names[i+1]} {np.median(d[:, i]):.2f}/{np.percentile(d[:, i], 90):.2f}/{d[:, i].max():.2f}" for i in range(5)))
life = (st[:, 0, 5] - st[:, 0, 0]) * tick
print(f"workgroup lifetime us: median {np.median(life):.2f} p90 {np.percentile(life, 90):.2f} max {life.max():.2f}; launch span {(st[:, :, 5].max() - t0) * tick:.2f} us")
# rounds per CU: group by (xcc, cu bits of HW_ID)
xcc = st[:, 0, 6]; hw = st[:, 0, 7]
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
key = xcc * 1000 + se * 100 + sh * 20 + cu
gaps = []
per = {}
for k in np.unique(key):
    rows = st[key == k]
    rows = rows[np.argsort(rows[:, 0, 0])]
    per[k] = len(rows)
    for a, b in zip(rows[:-1], rows[1:]):
        gaps.append((b[0, 0] - a[0, 5]) * tick)
gaps = np.array(gaps)
cnt = np.array(list(per.values()))
print(f"distinct CUs seen: {len(per)}; workgroups per CU min/median/max {cnt.min()}/{int(np.median(cnt))}/{cnt.max()}")
if len(gaps):
    print(f"gap between a workgroup's end stamp and the next workgroup's entry stamp on the same CU, us: median {np.median(gaps):.2f} p90 {np.percentile(gaps, 90):.2f} max {gaps.max():.2f}")
starts = np.sort((st[:, 0, 0] - t0) * tick)
print("entry times us (sorted), every 64th:", " ".join(f"{v:.1f}" for v in starts[::64]))
