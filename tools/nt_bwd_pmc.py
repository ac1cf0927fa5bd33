"""The backward's input-gradient products at B = 128 (M = 4 352 rows), as layers.gru_stack_bwd / engine.GeneratorEngine.backward issue them:
  (a) GRU input gradient  dx = [dgi_fwd | dgi_rev] @ [W_ih_fwd ; W_ih_rev]   [4352 x 600 x 1800], K-concatenated weight segments
  (b) text-encoder conv input gradient (two taps, dilation d) with the relu / dropout gate epilogue   [4352 x 300 x 600]
Target of rocprofv3 --pmc passes (tools/r5_pmc.sh) and, with --time, a same-process timing of both (HIP events, 200 launches each)."""
import importlib, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win, L = pkg.ops, pkg.ops.Win, pkg.layers
dev = torch.device("cuda:0")
nb, T, H = 128, 34, 300
M = nb * T
dgi = torch.randn(2, M, 3 * H, device=dev)
wt = torch.randn(2, 2 * H, 3 * H, device=dev) * 0.05            # the two transposed W_ih, [Kin][3H] each, one allocation (seg = Kin * 3H floats)
dx = torch.empty(M, 2 * H, device=dev)
a_cat = Win(dgi, batches=1, batch_stride=0, row_stride=3 * H, rows_in=2 * M, rows_out=M, cw=3 * H, K=6 * H, dil=M)
# round 6: the product as layers.gru_stack_bwd issues it now -- fp16 x 2 on the mover-wave kernel (128 x 96 tiles), planes of the K-concatenated
# transposed W_ih pair, dgi's magnitudes per clip (TG_GEMM_H2=0: the round-5 staged-slab form)
if ops.gemm_h2():
    w_f, w_r = wt[0].t().contiguous(), wt[1].t().contiguous()                       # the parameters: [3H][Kin]
    pl = ops.split2h_planes_tcat(w_f, w_r)
    clipmax = dgi.view(2, nb, T * 3 * H).abs().amax(dim=2).contiguous().view(-1)   # what tg_gru_backward_cluster_stats leaves
def gru_dx():
    if ops.gemm_h2():
        ops.gemm_nt(a_cat, wt[0], None, dx, b_seg=(3 * H, 2 * H * 3 * H), w_planes=pl, a_rowmax=clipmax, a_rowmax_rows=T)
    else:
        ops.gemm_nt(a_cat, wt[0], None, dx, b_seg=(3 * H, 2 * H * 3 * H))
d = 4
dc3 = torch.randn(nb, T, 300, device=dev)
wT = torch.randn(300, 600, device=dev) * 0.05
dh = torch.empty(M, 300, device=dev)
o0 = torch.randn(M, 300, device=dev)
m0 = (torch.rand(M, 300, device=dev) > 0.3).float() / 0.7
a_win = Win.taps(dc3, 2, shift=d, dil=-d, rows_out=T)
def tcn_dx():
    ops.gemm_nt(a_win, wT, None, dh, out_scale=m0, gate=o0)
n = 200 if "--time" in sys.argv else 6
for fn, name in ((gru_dx, "gru_dx [4352 x 600 x 1800]"), (tcn_dx, "tcn_dx [4352 x 300 x 600] + gate")):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    if "--time" in sys.argv:
        print(f"{name}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch (sustained, {n} launches)")
