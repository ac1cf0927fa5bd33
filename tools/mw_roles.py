"""Which role sets the pace of the mover-wave NT kernel?  Lab library: every role accumulates the shader-clock cycles it spends inside its
barriers (arrival -> release); per workgroup [waited, lifetime] for matrix wave 0 and mover wave 8.  2 x [13056 x 900 x 600], one launch after
warm-up, optional TG_MW_ABL ablation code as argv[1]."""
import ctypes as C, importlib, os, sys, torch
import numpy as np
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
pkg._lib.LIB_PATH = os.path.join(os.path.dirname(pkg._lib.LIB_PATH), "libtrimodal_hip_lab.so")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
shapes = {"gru": (13056, 900, 600, 2), "gru0": (13056, 900, 108, 2), "tcn": (13056, 300, 600, 1)}
M, N, K, n = shapes[sys.argv[2] if len(sys.argv) > 2 else "gru"]
x = torch.randn(M, K, device=dev)
ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(n)]
outs = [torch.empty(M, N, device=dev) for _ in range(n)]
if os.environ.get("TG_ABL_H2", "0") == "1":          # round 6: the fp16 x 2 instantiation
    sc = ops.h2_row_scales(Win.plain(x))
    probs = [dict(A=Win.plain(x), W=w, bias=None, out=o, w_planes=ops.split2h_planes(w), a_row_scale=sc) for w, o in zip(ws, outs)]
else:
    probs = [dict(A=Win.plain(x), W=w, bias=None, out=o, w_planes=ops.split3_planes(w)) for w, o in zip(ws, outs)]
os.environ["TG_MW_ABL"] = sys.argv[1] if len(sys.argv) > 1 else "0"
for _ in range(20):
    ops.gemm_nt_group(probs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm_nt_group(probs); e1.record(); torch.cuda.synchronize()
buf = (C.c_uint64 * (512 * 24))()
assert pkg._lib.load().tg_lab_mw_role_cycles(buf, 512 * 24) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(512, 12, 2).astype(np.float64)
st = st[st[:, 0, 1] > 0]
print(f"ABL={os.environ['TG_MW_ABL']} plan {ops.nt_kernel_plan(probs)} launch {e0.elapsed_time(e1) * 1e3:.1f} us, {len(st)} workgroups; per wave: lifetime cycles (s_memtime), share of it inside barriers")
for w in range(12):
    wt, tot = st[:, w, 0], st[:, w, 1]
    print(f"  wave {w:2d} ({'matrix' if w < 8 else 'mover '}): lifetime {np.median(tot):9.0f}  inside barriers {100 * np.median(wt / tot):5.1f} %  (p10 {100 * np.percentile(wt / tot, 10):.1f} %, p90 {100 * np.percentile(wt / tot, 90):.1f} %)")
