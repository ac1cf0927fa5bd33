"""Op-level parity: every C-ABI kernel against a plain torch fp64 CPU reference of the same op (GPU only).
Tolerances: fp32 kernels, max|err| / max|ref| <= 1e-5 forward, 1e-4 for long reductions (stated per test)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def cl(x):   # (B, C, L) -> channel-last (B, L, C)
    return x.transpose(1, 2).contiguous()


# ------------------------------------------------------------------------------------------------ GEMM family
@pytest.mark.parametrize("M,N,K", [(200, 70, 108), (37, 27, 150), (4352, 900, 600), (5, 1, 28), (130, 33, 8),
                                   (4352, 32, 960), (130, 20, 300), (33, 7, 264)])        # last three: narrow + long K -> the K-split variant
def test_gemm_nt_plain(pkg, dev, M, N, K):
    ops, Win = pkg.ops, pkg.ops.Win
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    ref = F.leaky_relu(x.double() @ w.double().t() + b.double(), 0.3)
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm_nt(Win.plain(x.to(dev)), w.to(dev), b.to(dev), out, act_slope=0.3)
    assert rel(out, ref) < 1e-5
    ops.gemm_nt(Win.plain(x.to(dev)), w.to(dev), None, out, accumulate=True)
    assert rel(out, ref + x.double() @ w.double().t()) < 1e-5


def test_gemm_nt_strided_views(pkg, dev):
    ops, Win = pkg.ops, pkg.ops.Win
    big = rnd(50, 108, seed=4).to(dev)
    w = rnd(12, 32, seed=5).to(dev)
    outbig = torch.zeros(50, 40, device=dev)
    ops.gemm_nt(Win.plain(big[:, 60:92]), w, None, outbig[:, 8:20])
    ref = big[:, 60:92].double().cpu() @ w.double().cpu().t()
    assert rel(outbig[:, 8:20], ref) < 1e-5 and float(outbig[:, :8].abs().max()) == 0 and float(outbig[:, 20:].abs().max()) == 0


@pytest.mark.parametrize("Ci,Co,kw,stride,pad,dil,L", [(16, 32, 15, 6, 0, 1, 211), (1, 16, 15, 5, 40, 1, 333), (27, 16, 3, 1, 0, 1, 34),
                                                      (300, 300, 2, 1, 4, 4, 34), (64, 64, 4, 2, 0, 1, 30)])
def test_conv_forward_wgrad_dgrad(pkg, dev, Ci, Co, kw, stride, pad, dil, L):
    Lm = pkg.layers
    B = 3
    x = rnd(B, Ci, L, seed=6).double().requires_grad_(True)
    w = rnd(Co, Ci, kw, seed=7, scale=0.2).double().requires_grad_(True)
    b = rnd(Co, seed=8).double().requires_grad_(True)
    y = F.conv1d(x, w, b, stride=stride, padding=pad, dilation=dil)
    causal = pad == dil * (kw - 1) and pad > 0
    if causal:
        y = y[:, :, :L]                          # Chomp1d
    dy = rnd(*y.shape, seed=9).double()
    y.backward(dy)
    xg, wg = cl(x.detach().float()).to(dev), w.detach().float().to(dev)
    out = Lm.conv_fwd(xg, Lm.pack_conv_weight(wg), b.detach().float().to(dev), kw, stride=stride, pad=pad, dil=dil,
                      rows_out=L if causal else None)
    assert rel(out, cl(y)) < 1e-5
    dW, db = torch.zeros_like(wg), torch.zeros(Co, device=dev)
    dyg = cl(dy.float()).to(dev)
    Lm.conv_wgrad(dyg, xg, dW, db, kw, stride=stride, pad=pad, dil=dil)
    assert rel(dW, w.grad) < 1e-5 and rel(db, b.grad) < 1e-5
    if pad == 0 and dil == 1:
        dx = Lm.conv_dgrad(dyg, wg, L, stride=stride)
        assert rel(dx, cl(x.grad)) < 1e-5


def test_conv_transpose_fwd_bwd(pkg, dev):
    Lm = pkg.layers
    B, Ci, Co, kw, L = 5, 4, 32, 3, 34
    x = rnd(B, Ci, L, seed=10).double().requires_grad_(True)
    w = rnd(Ci, Co, kw, seed=11, scale=0.3).double().requires_grad_(True)
    b = rnd(Co, seed=12).double().requires_grad_(True)
    y = F.conv_transpose1d(x, w, b)
    dy = rnd(*y.shape, seed=13).double()
    y.backward(dy)
    xg, wg = cl(x.detach().float()).to(dev), w.detach().float().to(dev)
    out = Lm.conv_transpose_fwd(xg, wg, b.detach().float().to(dev))
    assert rel(out, cl(y)) < 1e-5
    dW, db = torch.zeros_like(wg), torch.zeros(Co, device=dev)
    dx = Lm.conv_transpose_bwd(cl(dy.float()).to(dev), xg, wg, dW, db)
    assert rel(dW, w.grad) < 1e-5 and rel(db, b.grad) < 1e-5 and rel(dx, cl(x.grad)) < 1e-5


def test_linear_bwd_and_colsum(pkg, dev):
    Lm = pkg.layers
    M, N, K = 4352, 150, 300
    x, w, dy = rnd(M, K, seed=14), rnd(N, K, seed=15, scale=0.1), rnd(M, N, seed=16)
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    dx = Lm.linear_bwd(dy.to(dev), x.to(dev), w.to(dev), dW, db)
    assert rel(dW, dy.double().t() @ x.double()) < 1e-5
    assert rel(db, dy.double().sum(0)) < 1e-5 and rel(dx, dy.double() @ w.double()) < 1e-5


# ------------------------------------------------------------------------------------------------ GRU
@pytest.mark.parametrize("B,T,H,Kin", [(37, 9, 300, 108), (70, 28, 64, 8), (4, 34, 300, 600)])
def test_gru_layer_fwd_bwd(pkg, dev, B, T, H, Kin):
    Lm, ops = pkg.layers, pkg.ops
    gru = torch.nn.GRU(Kin, H, num_layers=1, batch_first=True, bidirectional=True).double()
    x = rnd(B, T, Kin, seed=17).double().requires_grad_(True)
    y, _ = gru(x)
    dy = rnd(B, T, 2 * H, seed=18).double()
    y.backward(dy)
    P = {f"gru.{k}": v.detach().float().to(dev).contiguous() for k, v in gru.named_parameters()}
    G = {k: torch.zeros_like(v) for k, v in P.items()}
    yg, tape = Lm.gru_stack_fwd(x.detach().float().to(dev), P, "gru", 1, H, p_drop=0.0, training=True, save=True)
    assert rel(yg, y) < 1e-5
    dx = Lm.gru_stack_bwd(dy.float().to(dev), tape, P, G, "gru", 1)
    assert rel(dx, x.grad) < 1e-4
    for k, v in gru.named_parameters():
        assert rel(G[f"gru.{k}"], v.grad) < 1e-4, k
    # sub-batch backward of a larger taped forward (the stacked-forward schedule of the GAN step)
    if B >= 8:
        b0, nb = 2, B // 2
        G2 = {k: torch.zeros_like(v) for k, v in P.items()}
        x2 = x.detach().clone().requires_grad_(True)
        y2, _ = gru(x2[b0:b0 + nb])
        gru.zero_grad()
        y2.backward(dy[b0:b0 + nb])
        dx2 = Lm.gru_stack_bwd(dy[b0:b0 + nb].float().to(dev).contiguous(), tape, P, G2, "gru", 1, b0=b0, nb=nb)
        assert rel(dx2, x2.grad[b0:b0 + nb]) < 1e-4
        for k, v in gru.named_parameters():
            assert rel(G2[f"gru.{k}"], v.grad) < 1e-4, k


@pytest.mark.parametrize("B", [3, 37, 128, 384, 768])
def test_gru_cluster_kernels_match_step_launches(pkg, dev, B):
    """The persistent cluster-synchronised recurrence (csrc/gru_cluster.hip) against the per-step launches (csrc/gru.hip): same
    K-slicing and summation order, so they agree to rounding; a stale inter-workgroup hand-off would show as an O(1e-2) error.
    Repeated on the same buffers (the L2s then hold the previous run's lines of the exchange buffer).
    (A hand-off soak, not the kernels' parity test: that comes from the B = 128 golden / trajectory tests, which assert the cluster kernels
    ran.)  B = 768 = the stacked forward of --batch 256: two row chunks of 384 on one workspace (ops.gru_cluster_chunks), backward of a
    256-row group as two chunks of 128."""
    ops = pkg.ops
    T, H = 34, 300
    g = torch.Generator().manual_seed(B)
    w = [(torch.randn(3 * H, H, generator=g) * 0.08).to(dev) for _ in range(2)]
    b = [(torch.randn(3 * H, generator=g) * 0.05).to(dev) for _ in range(2)]
    wt = [x.t().contiguous() for x in w]
    prev = ops.GRU_CLUSTER
    try:
        for rep in range(4):
            gi = (torch.randn(2, B, T, 3 * H, generator=g) * 0.5).to(dev)
            dy = torch.randn(B, T, 2 * H, generator=g).to(dev)
            out = {}
            for cluster in (False, True):
                ops.GRU_CLUSTER = cluster
                y = torch.full((B, T, 2 * H), float("nan"), device=dev)
                sv = torch.full((2, B, T, 4 * H), float("nan"), device=dev)
                ops.gru_forward(gi, w, b, y, sv)
                nb = min(B, 128) if B < 768 else 256               # backward of one group of the stacked forward
                b0 = (B - nb) // 2
                dgi = torch.full((2, nb, T, 3 * H), float("nan"), device=dev)
                dgh = torch.full((2, nb, T, 3 * H), float("nan"), device=dev)
                ops.gru_backward(dy[b0:b0 + nb].contiguous(), y, sv, wt, dgi, dgh, torch.zeros(4 * nb * H, device=dev), b0=b0, nb=nb)
                out[cluster] = (y, sv, dgi, dgh)
            ops.check_async_errors()
            for a, c in zip(out[False], out[True]):
                assert bool(torch.isfinite(c).all())
                assert float((a - c).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max())), (B, rep)
        if B == 768:
            assert ops.gru_cluster_chunks(768, H) == [(0, 384), (384, 384)] and ops.gru_cluster_chunks(256, H, bwd=True) == [(0, 128), (128, 128)]
    finally:
        ops.GRU_CLUSTER = prev


def test_gru_cluster_saves_gates_for_requested_rows_only(pkg, dev):
    """tg_gru_forward_cluster_rows: of a stacked forward only one call is differentiated -- the gates are written for its batch rows, the
    rest of `save` is left untouched, y is unchanged (B = 48 and 384: one and two 16-row tiles per cluster)."""
    ops = pkg.ops
    T, H = 6, 300
    for B, r0, rn in ((48, 16, 16), (384, 128, 128), (40, 7, 13)):
        if not pkg._lib.load().tg_gru_cluster_supported(B, H):
            continue
        g = torch.Generator().manual_seed(B)
        w = [(torch.randn(3 * H, H, generator=g) * 0.08).to(dev) for _ in range(2)]
        b = [(torch.randn(3 * H, generator=g) * 0.05).to(dev) for _ in range(2)]
        gi = (torch.randn(2, B, T, 3 * H, generator=g) * 0.5).to(dev)
        y0, sv0 = torch.empty(B, T, 2 * H, device=dev), torch.empty(2, B, T, 4 * H, device=dev)
        ops.gru_forward(gi, w, b, y0, sv0)
        y1 = torch.empty_like(y0)
        sv1 = torch.full_like(sv0, float("nan"))
        ops.gru_forward(gi, w, b, y1, sv1, save_rows=(r0, rn))
        ops.check_async_errors()
        assert torch.equal(y0, y1) and torch.equal(sv1[:, r0:r0 + rn], sv0[:, r0:r0 + rn])
        assert bool(torch.isnan(sv1[:, :r0]).all()) and bool(torch.isnan(sv1[:, r0 + rn:]).all())


def test_gru_cluster_flag_generations_across_sequence_lengths(pkg, dev):
    """The cluster kernels never zero their flag words: each launch numbers them from the cluster's generation word and advances it by
    T + 1 (csrc/gru_cluster_x3.hip).  Launches of DIFFERENT lengths -- including T = 1 (nothing published) and T = 2 (the shortest with a
    hand-off) -- on the SAME workspace, back to back without a host sync in between, must each match the per-step launches: a stale flag
    accepted as current would hand a consumer the previous launch's h tile."""
    ops = pkg.ops
    H, B = 300, 40
    g = torch.Generator().manual_seed(5)
    w = [(torch.randn(3 * H, H, generator=g) * 0.08).to(dev) for _ in range(2)]
    b = [(torch.randn(3 * H, generator=g) * 0.05).to(dev) for _ in range(2)]
    wt = [x.t().contiguous() for x in w]
    prev = ops.GRU_CLUSTER
    try:
        jobs = []
        for T in (34, 2, 1, 7, 1, 34, 3, 2):
            gi = (torch.randn(2, B, T, 3 * H, generator=g) * 0.5).to(dev)
            dy = torch.randn(B, T, 2 * H, generator=g).to(dev)
            jobs.append((T, gi, dy))
        res = {}
        for cluster in (True, False):
            ops.GRU_CLUSTER = cluster
            outs = []
            for T, gi, dy in jobs:                                  # enqueued back to back
                y = torch.full((B, T, 2 * H), float("nan"), device=dev)
                sv = torch.full((2, B, T, 4 * H), float("nan"), device=dev)
                dgi = torch.full((2, B, T, 3 * H), float("nan"), device=dev)
                dgh = torch.full((2, B, T, 3 * H), float("nan"), device=dev)
                ops.gru_forward(gi, w, b, y, sv)
                ops.gru_backward(dy, y, sv, wt, dgi, dgh, torch.zeros(4 * B * H, device=dev))
                outs.append((y, sv, dgi, dgh))
            res[cluster] = outs
        ops.check_async_errors()
        for (T, _, _), oc, os_ in zip(jobs, res[True], res[False]):
            for a, c in zip(os_, oc):
                assert bool(torch.isfinite(c).all()), T
                assert float((a - c).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max())), T
    finally:
        ops.GRU_CLUSTER = prev


# ------------------------------------------------------------------------------------------------ BatchNorm
@pytest.mark.parametrize("rows,C,groups,slope", [(3 * 500, 16, 3, 0.3), (64, 256, 1, 1.0), (2 * 96, 8, 2, 1.0),
                                                  (3 * 2048, 8, 3, 1.0), (2 * 600, 64, 2, 0.2),     # two-launch kernels (bn2_*), several groups
                                                  (2 * 20001, 32, 2, 0.3),       # two-launch kernels, many partial workgroups
                                                  (50001, 12, 1, 0.3)])          # streaming kernels, scalar path (12 does not divide 1024)
def test_batchnorm_train_eval_backward(pkg, dev, rows, C, groups, slope):
    Lm = pkg.layers
    x = (rnd(rows, C, seed=19) * 2 + 0.5)
    gamma, beta = 1 + 0.1 * rnd(C, seed=20), 0.1 * rnd(C, seed=21)
    rm, rv = 0.05 * rnd(C, seed=22), 1 + 0.1 * rnd(C, seed=23).abs()
    per = rows // groups
    # reference: `groups` separate nn.BatchNorm1d calls
    bn = torch.nn.BatchNorm1d(C).double()
    bn.weight.data, bn.bias.data = gamma.double().clone(), beta.double().clone()
    bn.running_mean.data, bn.running_var.data = rm.double().clone(), rv.double().clone()
    xs = x.double().requires_grad_(True)
    ys = torch.cat([F.leaky_relu(bn(xs[g * per:(g + 1) * per]), slope) for g in range(groups)])
    dy = rnd(rows, C, seed=24).double()
    g_sel = groups - 1
    ys[g_sel * per:(g_sel + 1) * per].backward(dy[g_sel * per:(g_sel + 1) * per])
    rmg, rvg, nbt = rm.to(dev).clone(), rv.to(dev).clone(), torch.zeros((), dtype=torch.int64, device=dev)
    y, st = Lm.bn_fwd(x.to(dev), gamma.to(dev), beta.to(dev), rmg, rvg, nbt, training=True, groups=groups, act_slope=slope)
    assert rel(y, ys) < 1e-5 and rel(rmg, bn.running_mean) < 1e-5 and rel(rvg, bn.running_var) < 1e-5 and int(nbt) == groups
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dx = Lm.bn_bwd(dy[g_sel * per:(g_sel + 1) * per].float().to(dev), st, gamma.to(dev), beta.to(dev), dg, db, g0=g_sel, ng=1,
                   row0=g_sel * per)
    assert rel(dx, xs.grad[g_sel * per:(g_sel + 1) * per]) < 1e-4 and rel(dg, bn.weight.grad) < 1e-4 and rel(db, bn.bias.grad) < 1e-4
    if groups > 1:                                   # all groups of the stacked call at once (the discriminator step's real + fake halves)
        xs2 = x.double().requires_grad_(True)
        bn2m = torch.nn.BatchNorm1d(C).double()
        bn2m.weight.data, bn2m.bias.data = gamma.double().clone(), beta.double().clone()
        ys2 = torch.cat([F.leaky_relu(bn2m(xs2[g * per:(g + 1) * per]), slope) for g in range(groups)])
        ys2.backward(dy)
        dg2, db2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dx2 = Lm.bn_bwd(dy.float().to(dev), st, gamma.to(dev), beta.to(dev), dg2, db2, g0=0, ng=groups, row0=0)
        assert rel(dx2, xs2.grad) < 1e-4 and rel(dg2, bn2m.weight.grad) < 1e-4 and rel(db2, bn2m.bias.grad) < 1e-4
    bn.eval()
    ye, _ = Lm.bn_fwd(x.to(dev), gamma.to(dev), beta.to(dev), rmg, rvg, nbt, training=False, act_slope=slope)
    assert rel(ye, F.leaky_relu(bn(x.double()), slope)) < 1e-5
    # repeats: the same batch normalised by three identical calls
    rm3, rv3, nbt3 = rm.to(dev).clone(), rv.to(dev).clone(), torch.zeros((), dtype=torch.int64, device=dev)
    bn3 = torch.nn.BatchNorm1d(C).double(); bn3.running_mean.data, bn3.running_var.data = rm.double().clone(), rv.double().clone()
    for _ in range(3):
        bn3(x[:per].double())
    Lm.bn_fwd(x[:per].to(dev).contiguous(), gamma.to(dev), beta.to(dev), rm3, rv3, nbt3, training=True, repeats=3)
    assert rel(rm3, bn3.running_mean) < 1e-5 and rel(rv3, bn3.running_var) < 1e-5 and int(nbt3) == 3


# ------------------------------------------------------------------------------------------------ element-wise & co
def test_elementwise_family(pkg, dev):
    ops = pkg.ops
    a, b = rnd(1000, seed=25).to(dev), rnd(1000, seed=26).to(dev)
    assert rel(ops.add_relu(a, b, torch.empty_like(a)), torch.relu(a.cpu() + b.cpu())) == 0
    y = torch.relu(a)
    m = (rnd(1000, seed=27) > 0).float().to(dev) * 1.25
    assert rel(ops.act_mask_bwd(b, y, m, 0.0, torch.empty_like(a)), b.cpu() * m.cpu() * (y.cpu() > 0)) == 0
    assert rel(ops.mul(a, m, torch.empty_like(a)), a.cpu() * m.cpu()) == 0
    c = b.clone(); ops.axpy(a, c, 0.5, accumulate=True); assert rel(c, b.cpu() + 0.5 * a.cpu()) < 1e-6
    src = rnd(20, 7, seed=28).to(dev); dst = torch.zeros(20, 11, device=dev)
    ops.copy2d(src, dst[:, 2:9]); assert torch.equal(dst[:, 2:9].cpu(), src.cpu()) and float(dst[:, 9:].abs().max()) == 0
    z = rnd(5, 16, seed=29).to(dev); big = torch.zeros(5 * 6, 20, device=dev)
    ops.repeat_rows(z, big[:, 4:], 5, 6)
    assert torch.equal(big.view(5, 6, 20)[:, :, 4:].cpu(), z.cpu()[:, None, :].expand(5, 6, 16))
    s = ops.sum_rows(big[:, 4:], torch.empty(5, 16, device=dev), 5, 6); assert rel(s, 6 * z.cpu()) < 1e-6
    yy = rnd(12, 10, seed=30).to(dev)
    assert rel(ops.add_halves(yy, torch.empty(12, 5, device=dev)), yy.cpu()[:, :5] + yy.cpu()[:, 5:]) == 0
    dd = ops.dup_halves(yy[:, :5].contiguous(), torch.empty(12, 10, device=dev))
    assert torch.equal(dd[:, :5].cpu(), dd[:, 5:].cpu()) and torch.equal(dd[:, :5].cpu(), yy[:, :5].cpu())
    t = rnd(3, 5, 7, seed=31).to(dev)
    for perm in ((0, 2, 1), (2, 1, 0), (1, 2, 0)):
        assert torch.equal(ops.permute3(t, torch.empty(t.numel(), device=dev), perm).view([t.shape[p] for p in perm]).cpu(),
                           t.cpu().permute(*perm).contiguous())
    # batched jobs (layers.WeightPrep's table): inner-dimension swaps take the LDS-tiled path, anything else the element-wise one; ragged
    # sizes, one workgroup range per job
    srcs = [rnd(1, 900, 300, seed=33).to(dev), rnd(4, 37, 70, seed=34).to(dev), rnd(16, 15, 2, seed=35).to(dev), rnd(3, 5, 7, seed=36).to(dev)]
    perms = [(0, 2, 1), (0, 2, 1), (0, 2, 1), (2, 0, 1)]
    dsts = [torch.full((x.numel(),), float("nan"), device=dev) for x in srcs]
    rows, wg0 = [], 0
    for x, d_, pm in zip(srcs, dsts, perms):
        nwg = max(1, min(64, (x.numel() + 2047) // 2048))
        rows.append([x.data_ptr(), d_.data_ptr(), *x.shape, *pm, wg0, nwg])
        wg0 += nwg
    ops.permute3_batch(torch.tensor(rows, dtype=torch.int64).to(dev), len(rows), wg0)
    for x, d_, pm in zip(srcs, dsts, perms):
        assert torch.equal(d_.view([x.shape[p] for p in pm]).cpu(), x.cpu().permute(*pm).contiguous()), (tuple(x.shape), pm)
    tg = rnd(6, 34, 27, seed=32).to(dev)
    pre = ops.make_pre_seq(tg, torch.empty(6, 34, 28, device=dev), 4).cpu()
    assert torch.equal(pre[:, :4, :27], tg.cpu()[:, :4]) and float(pre[:, 4:].abs().max()) == 0 and torch.all(pre[:, :4, 27] == 1)
    sg = ops.sigmoid(a, torch.empty_like(a)); assert rel(sg, torch.sigmoid(a.cpu().double())) < 1e-6
    assert rel(ops.sigmoid_bwd(b, sg, torch.empty_like(a)), b.cpu().double() * sg.cpu().double() * (1 - sg.cpu().double())) < 1e-6


def test_residual_block_gates_one_pass(pkg, dev):
    ops = pkg.ops
    for n in (4096, 1003):                                       # 16-byte and scalar paths
        dy, y, o = rnd(n, seed=81).to(dev), rnd(n, seed=82).to(dev), rnd(n, seed=83).to(dev)
        m = ((torch.rand(n, generator=torch.Generator().manual_seed(9)) > 0.3).float() / 0.7).to(dev)
        dsum, dc = ops.act_mask_bwd2(dy, y, o, m, 0.0, torch.empty_like(dy), torch.empty_like(dy))
        ref_s = dy * (y > 0).float()
        assert torch.equal(dsum, ref_s) and torch.equal(dc, ref_s * (o > 0).float() * m)
        dsum2, dc2 = ops.act_mask_bwd2(dy, y, o, None, 0.2, torch.empty_like(dy), torch.empty_like(dy))
        assert torch.equal(dc2, ref_s * torch.where(o > 0, torch.ones_like(o), torch.full_like(o, 0.2)))
        assert torch.equal(ops.act_mask_bwd(dy, y, m, 0.0, torch.empty_like(dy)), ref_s * m)
        assert torch.equal(ops.mul(dy, m, torch.empty_like(dy)), dy * m)
        assert torch.equal(ops.add_relu(dy, y, torch.empty_like(dy)), torch.relu(dy + y))


def test_embedding_gather_scatter(pkg, dev):
    ops = pkg.ops
    V, D, n = 50, 300, 4 * 34
    table = rnd(V, D, seed=33)
    idx = torch.zeros(n, dtype=torch.int64); idx[::7] = torch.randint(4, V, (len(idx[::7]),), generator=torch.Generator().manual_seed(1))
    out = ops.embed_gather(table.to(dev), idx.to(dev), torch.empty(n, D, device=dev))
    assert torch.equal(out.cpu(), table[idx])
    dout = rnd(n, D, seed=34)
    dt = ops.embed_scatter_add(dout.to(dev), idx.to(dev), torch.zeros(V, D, device=dev))
    ref = torch.zeros(V, D, dtype=torch.float64).index_add_(0, idx, dout.double())
    assert rel(dt, ref) < 1e-5


def test_deterministic_forms_match_the_default_ones(pkg, dev):
    """tg_set_deterministic(1) swaps the float-atomic combines for fixed-order ones (csrc/elementwise.hip embed_scatter_det_kernel: sixteen
    waves per first occurrence, partial rows added in wave order; csrc/losses.hip: the head's parameter gradients by one workgroup).  The
    bit-identity test in test_trajectory_gpu.py only says that they are REPRODUCIBLE; here they are held to the same fp64 references as the
    default forms -- a word batch's index pattern (a padding id that occurs thousands of times, ids that occur once), accumulation into a
    non-zero table -- and run twice for bit-identity."""
    ops = pkg.ops
    g = torch.Generator().manual_seed(77)
    V, D, n = 2000, 300, 128 * 34
    idx = torch.zeros(n, dtype=torch.int64)
    hot = torch.randperm(n, generator=g)[:1300]
    idx[hot] = torch.randint(4, V, (1300,), generator=g)
    idx[5] = V - 1; idx[n - 1] = 7; idx[0] = 7
    dout = torch.randn(n, D, generator=g)
    base = torch.randn(V, D, generator=g)
    ref = base.double().index_add_(0, idx, dout.double())
    B, T, H = 256, 28, 64
    y, l1 = torch.randn(B, T, 2 * H, generator=g).to(dev), torch.randn(B, T, generator=g).to(dev)
    w1, w2 = torch.randn(H, generator=g).to(dev), torch.randn(T, generator=g).to(dev)
    dl = torch.randn(B, generator=g).to(dev)
    outs = {}
    try:
        for det in (False, True, True):
            ops.set_deterministic(det)
            dt = ops.embed_scatter_add(dout.to(dev), idx.to(dev), base.clone().to(dev))
            assert rel(dt, ref) < 1e-5, det
            gr = [torch.zeros(H, device=dev), torch.zeros(1, device=dev), torch.zeros(T, device=dev), torch.zeros(1, device=dev)]
            dy = ops.d_head_bwd(dl, y, l1, w1, w2, torch.empty_like(y), gr)
            if det and True in outs:
                assert torch.equal(outs[True][0], dt) and all(torch.equal(a, b) for a, b in zip(outs[True][1], gr)) and torch.equal(outs[True][2], dy)
            outs[det] = (dt, gr, dy)
    finally:
        ops.set_deterministic(False)
    assert torch.equal(outs[False][2], outs[True][2])                       # the input gradient has no cross-workgroup sum: same bits
    for a, b in zip(outs[False][1], outs[True][1]):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()), (a, b)
    # the head's parameter gradients against fp64
    ys = (y[:, :, :H] + y[:, :, H:]).double()
    dl1 = dl.double()[:, None] * w2.double()[None, :]
    assert rel(outs[True][1][0], (dl1[:, :, None] * ys).sum((0, 1))) < 1e-5 and rel(outs[True][1][2], (dl.double()[:, None] * l1.double()).sum(0)) < 1e-5
    assert rel(outs[True][1][1], dl1.sum().reshape(1)) < 1e-5 and rel(outs[True][1][3], dl.double().sum().reshape(1)) < 1e-5
    # column sums (bias gradients outside the products): both workgroup shapes of the deterministic kernel, accumulate and overwrite
    for M, N in ((4352, 900), (7168, 192), (300, 27), (5000, 4160)):
        X = torch.randn(M, N + 3, generator=g).to(dev)[:, :N]
        acc0 = torch.randn(N, generator=g).to(dev)
        res = {}
        for det in (False, True, True):
            ops.set_deterministic(det)
            try:
                a = ops.colsum(X, acc0.clone(), accumulate=True)
                b = ops.colsum(X, torch.full((N,), 7.0, device=dev), accumulate=False)
            finally:
                ops.set_deterministic(False)
            assert rel(a, acc0.double() + X.double().sum(0)) < 1e-5 and rel(b, X.double().sum(0)) < 1e-5, (M, N, det)
            if det and True in res:
                assert torch.equal(res[True][0], a) and torch.equal(res[True][1], b)
            res[det] = (a, b)


@pytest.mark.parametrize("nb,step", [(128, "d"), (5, "d"), (1, "d"), (128, "g"), (7, "g")])
def test_head_and_gan_loss_terms_in_one_launch(pkg, dev, nb, step):
    """tg_d_head_step (head forward + per-clip GAN loss terms + head backward, multimodal_context_net.py:243-252 under train_gan.py:36-41
    for the discriminator step's stacked [real ; fake] batch and :55-57,86-88 for the generator step) against fp64 autograd of the same
    expression and against the three launches it replaces; twice on the same arrival counter (the kernel must leave it at zero), with and
    without the in-kernel loss scalar, and in deterministic mode (parameter gradients by the fixed-order kernel)."""
    ops = pkg.ops
    g = torch.Generator().manual_seed(900 + nb)
    T, H, w_gan = 28, 64, 5.0
    n_rows = 2 * nb if step == "d" else nb
    y = torch.randn(n_rows, T, 2 * H, generator=g)
    w1, b1 = torch.randn(H, generator=g) * 0.2, torch.randn(1, generator=g)
    w2, b2 = torch.randn(T, generator=g) * 0.2, torch.randn(1, generator=g)
    yd, pr = y.double().requires_grad_(True), [t.double().requires_grad_(True) for t in (w1, b1, w2, b2)]
    l1_ref = (yd[:, :, :H] + yd[:, :, H:]) @ pr[0] + pr[1]
    logit_ref = l1_ref @ pr[2] + pr[3]
    p_ref = torch.sigmoid(logit_ref)
    if step == "d":
        loss_ref = -torch.mean(torch.log(p_ref[:nb] + 1e-8) + torch.log(1 - p_ref[nb:] + 1e-8))          # dis_error (:41)
        loss_ref.backward()
        scales = (1.0 / nb, 1.0 / nb)
    else:
        loss_ref = -torch.mean(torch.log(p_ref + 1e-8))                                                  # gen_error (:57), weighted in the total (:86)
        (w_gan * loss_ref).backward()
        scales = (w_gan / nb, 0.0)
    loss_ref = float(loss_ref.detach())
    yg, (w1g, b1g, w2g, b2g) = y.to(dev), [t.to(dev) for t in (w1, b1, w2, b2)]
    l1_s, logit_s, prob_s = ops.d_head_fwd(yg, w1g, b1g, w2g, b2g)                                       # the separate launches
    if step == "d":
        out_s, dl_s = torch.empty(1, device=dev), torch.empty(2 * nb, device=dev)
        ops.gan_d_loss(logit_s.view(-1)[:nb], logit_s.view(-1)[nb:], out_s, dl_s[:nb], dl_s[nb:])
        assert abs(float(out_s) - loss_ref) <= 1e-5 * max(1.0, abs(loss_ref))
    runs = []
    try:
        for det in (False, False, True, True):
            ops.set_deterministic(det)
            base = [torch.full((H,), 0.5, device=dev), torch.full((1,), -1.0, device=dev), torch.zeros(T, device=dev), torch.zeros(1, device=dev)]
            gr = [t.clone() for t in base] if step == "d" else None
            out = torch.full((1,), 7.0, device=dev)
            o = ops.d_head_step(yg, w1g, b1g, w2g, b2g, nb, scales[0], scales[1], out, gr)
            assert rel(o["l1"], l1_ref.detach()) < 1e-5 and rel(o["logit"].view(-1), logit_ref.detach()) < 1e-5 and rel(o["prob"].view(-1), p_ref.detach()) < 1e-5
            assert abs(float(out) - loss_ref) <= 1e-5 * max(1.0, abs(loss_ref)), (float(out), loss_ref, det)
            assert abs(-math.fsum(o["terms"].tolist()) / nb - loss_ref) <= 1e-5 * max(1.0, abs(loss_ref))
            assert rel(o["dy"], yd.grad) < 1e-5
            assert rel(o["l1"], l1_s.double()) < 1e-5
            if step == "d":
                for got, b0, ref in zip(gr, base, (t.grad for t in pr)):
                    assert rel(got - b0, ref.reshape(got.shape)) < 2e-5, det       # accumulated onto what the gradient slab held
                assert rel(o["d_logit"], dl_s.double()) < 1e-5
            o2 = ops.d_head_step(yg, w1g, b1g, w2g, b2g, nb, scales[0], scales[1], None, None)     # no scalar, input gradient only
            assert all(torch.equal(o[k], o2[k]) for k in o)
            runs.append((o, gr, out))
    finally:
        ops.set_deterministic(False)
    for a, b in ((runs[0], runs[1]), (runs[2], runs[3])):
        assert all(torch.equal(a[0][k], b[0][k]) for k in a[0]) and torch.equal(a[2], b[2])      # (also: the counter was left at zero)
    if step == "d":
        assert all(torch.equal(x, z) for x, z in zip(runs[2][1], runs[3][1]))                     # deterministic mode: same bits twice
    assert torch.equal(runs[0][0]["dy"], runs[2][0]["dy"])


def test_weight_norm_and_dgrad_pack(pkg, dev):
    ops = pkg.ops
    Co, Ci, kw = 300, 300, 2
    v = rnd(Co, Ci, kw, seed=35, scale=0.05).double().requires_grad_(True)
    g = (1 + 0.1 * rnd(Co, 1, 1, seed=36)).double().requires_grad_(True)
    w = v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))
    dw = rnd(Co, Ci, kw, seed=37).double()
    w.backward(dw)
    vg, gg = v.detach().float().to(dev), g.detach().float().view(-1).to(dev)
    wp = ops.weight_norm_fwd(vg, gg, torch.empty(Co, kw * Ci, device=dev))
    assert rel(wp.view(Co, kw, Ci), w.detach().permute(0, 2, 1)) < 1e-6
    dg, dv = torch.zeros(Co, device=dev), torch.zeros_like(vg)
    ops.weight_norm_bwd(dw.permute(0, 2, 1).contiguous().float().to(dev).view(Co, kw * Ci), vg, gg, dg, dv)
    assert rel(dg, g.grad.view(-1)) < 1e-5 and rel(dv, v.grad) < 1e-5
    # the batched launch == the single one, per conv (three convs with different data, accumulating into non-zero gradients)
    dwp = dw.permute(0, 2, 1).contiguous().float().to(dev).view(Co, kw * Ci)
    vs = [vg * (1 + 0.1 * i) for i in range(3)]
    gs = [gg + 0.01 * i for i in range(3)]
    dws = [dwp * (1 - 0.2 * i) for i in range(3)]
    dgs_b, dvs_b = [torch.full((Co,), 0.5, device=dev) for _ in range(3)], [torch.full_like(vg, 0.25) for _ in range(3)]
    ops.weight_norm_bwd_batch(dws, vs, gs, dgs_b, dvs_b)
    for i in range(3):
        dg1, dv1 = torch.full((Co,), 0.5, device=dev), torch.full_like(vg, 0.25)
        ops.weight_norm_bwd(dws[i], vs[i], gs[i], dg1, dv1)
        assert torch.equal(dg1, dgs_b[i]) and torch.equal(dv1, dvs_b[i])
    # the batched forward (16-channel LDS tiles; Co = 300 leaves a ragged last tile): packed rows and the transposed pack, per conv
    wp_b, wt_b = ops.weight_norm_fwd_batch(vs, gs)
    for i in range(3):
        wi = vs[i].double() * (gs[i].double().view(-1, 1, 1) / vs[i].double().flatten(1).norm(dim=1).view(-1, 1, 1))
        assert rel(wp_b[i].view(Co, kw, Ci), wi.permute(0, 2, 1)) < 1e-6
        assert rel(wt_b[i].view(Ci, kw, Co), wi.permute(1, 2, 0)) < 1e-6
        assert torch.equal(wt_b[i].view(Ci, kw, Co), wp_b[i].view(Co, kw, Ci).permute(2, 1, 0))      # the two packs hold the same numbers
    wp_n, wt_n = ops.weight_norm_fwd_batch(vs[:1], gs[:1], want_t=False)
    assert wt_n is None and torch.equal(wp_n[0], wp_b[0])
    w2 = rnd(8, 5, 15, seed=38).to(dev)
    packed = ops.conv_dgrad_pack(w2, torch.empty(6, 5, 3 * 8, device=dev), 6).cpu().view(6, 5, 3, 8)
    for r in range(6):
        for j in range(3):
            k = r + 6 * j
            exp = w2.cpu()[:, :, k].t() if k < 15 else torch.zeros(5, 8)
            assert torch.equal(packed[r, :, j, :], exp)


def test_rng_ops(pkg, dev):
    ops = pkg.ops
    st = ops.new_rng_state(1234, dev)
    m = ops.dropout_mask(torch.empty(1 << 20, device=dev), 0.3, st, 1)
    keep = float((m > 0).float().mean())
    assert abs(keep - 0.7) < 3e-3 and abs(float(m.max()) - 1 / 0.7) < 1e-6 and float(m.min()) == 0
    m2 = ops.dropout_mask(torch.empty(1 << 20, device=dev), 0.3, st, 1)
    assert torch.equal(m, m2)                                     # same (seed, step, site) -> same draw
    ops.rng_advance(st)
    m3 = ops.dropout_mask(torch.empty(1 << 20, device=dev), 0.3, st, 1)
    assert not torch.equal(m, m3) and int(st[1]) == 1
    m4 = ops.dropout_mask(torch.empty(1 << 20, device=dev), 0.3, st, 2)
    assert abs(float(((m3 > 0) == (m4 > 0)).float().mean()) - (0.49 + 0.09)) < 5e-3     # sites are independent
    x = torch.randn(1000003, device=dev)                          # fused draw + apply == separate draw, then multiply
    y5, m5 = ops.dropout_apply(x, 0.3, st, 1)
    assert torch.equal(m5, ops.dropout_mask(torch.empty_like(x), 0.3, st, 1)) and torch.equal(y5, x * m5)
    assert torch.equal(m5, m3[:1000003])                          # draws do not depend on the tensor length
    e = ops.normal(torch.empty(1 << 20, device=dev), st, 3)
    assert abs(float(e.mean())) < 5e-3 and abs(float(e.std()) - 1) < 5e-3 and abs(float((e ** 4).mean()) - 3) < 0.1
    p = ops.randperm(torch.empty(128, dtype=torch.int64, device=dev), st, 4).cpu()
    assert sorted(p.tolist()) == list(range(128)) and p.tolist() != list(range(128))
    src = torch.arange(100, 228, device=dev)
    assert torch.equal(ops.gather_i64(src, p.to(dev), torch.empty_like(src)).cpu(), src.cpu()[p])


def test_reparam_losses_adam_vs_oracle(pkg, dev):
    from oracle import ref_model as O
    ops = pkg.ops
    B, T, D, Z = 6, 34, 27, 16
    mu, lv, eps = rnd(B, Z, seed=40).double().requires_grad_(True), (0.3 * rnd(B, Z, seed=41)).double().requires_grad_(True), rnd(B, Z, seed=42).double()
    z = mu + eps * torch.exp(0.5 * lv)
    zg = ops.reparam_fwd(mu.detach().float().to(dev), lv.detach().float().to(dev), eps.float().to(dev), torch.empty(B, Z, device=dev))
    assert rel(zg, z) < 1e-6
    out = (0.2 * rnd(B, T, D, seed=43)).double().requires_grad_(True)
    tgt, outr = 0.2 * rnd(B, T, D, seed=44).double(), 0.2 * rnd(B, T, D, seed=45).double()
    zr = rnd(B, Z, seed=46).double()
    zr[0] = z.detach()[0] + 1e-9                                   # forces the clamp(min=-1000) branch for clip 0
    logit = rnd(B, 1, seed=47).double().requires_grad_(True)
    for epoch in (0, 11):
        for t in (mu, lv, out, logit):
            t.grad = None
        loss, parts = O.gan_losses_g(out, tgt, torch.sigmoid(logit), outr, z, zr, mu, lv, epoch)
        # z enters the loss only detached (train_gan.py:71); mu/logvar via KLD
        loss.backward()
        f = lambda t: t.detach().float().to(dev).contiguous()
        sc, d_out, d_mu, d_lv, d_lg = (torch.empty(5, device=dev), torch.empty(B, T, D, device=dev), torch.empty(B, Z, device=dev),
                                       torch.empty(B, Z, device=dev), torch.empty(B, device=dev))
        ops.gan_g_loss(f(out), f(tgt), f(outr), f(z), f(zr), f(mu), f(lv), f(logit).view(-1), (500.0, 0.1, 0.05, 5.0), epoch > 10,
                       torch.empty(3 * B, device=dev), sc, d_out, d_mu, d_lv, d_lg)
        ref_sc = torch.stack([parts["huber"], parts["kld"], parts["div_reg"], parts["gen"], loss]).detach()
        assert rel(sc, ref_sc) < 2e-5, (sc.cpu(), ref_sc)
        assert float(parts["div_reg"]) < -100          # clamp branch really hit
        assert rel(d_out, out.grad) < 2e-5 and rel(d_mu, mu.grad) < 2e-5 and rel(d_lv, lv.grad) < 2e-5
        if epoch > 10:
            assert rel(d_lg, logit.grad.view(-1)) < 2e-5
        else:
            assert float(d_lg.abs().max()) == 0
    lr_, lf_ = rnd(B, seed=48).double().requires_grad_(True), rnd(B, seed=49).double().requires_grad_(True)
    dis = torch.sum(-torch.mean(torch.log(torch.sigmoid(lr_) + 1e-8) + torch.log(1 - torch.sigmoid(lf_) + 1e-8)))
    dis.backward()
    o, d1, d2 = torch.empty(1, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)
    ops.gan_d_loss(lr_.detach().float().to(dev), lf_.detach().float().to(dev), o, d1, d2)
    assert rel(o, dis.detach().view(1)) < 1e-5 and rel(d1, lr_.grad) < 1e-5 and rel(d2, lf_.grad) < 1e-5
    # AE loss
    rc = (0.3 * rnd(B, T, D, seed=50)).double().requires_grad_(True)
    l = O.ae_loss(rc, tgt); l.backward()
    o2, drc = torch.empty(1, device=dev), torch.empty(B, T, D, device=dev)
    ops.ae_loss(rc.detach().float().to(dev), tgt.float().to(dev), o2, drc)
    assert rel(o2, l.detach().view(1)) < 1e-5 and rel(drc, rc.grad) < 1e-5
    assert rel(ops.l1_mean(zg, zg * 0 + 1, torch.empty(1, device=dev)), (z.detach() - 1).abs().mean().view(1)) < 1e-5
    # Adam: three steps against the oracle's restatement of torch.optim.Adam
    n = 1003
    p = {"w": rnd(n, seed=51).double()}
    pg, mg, vg, step = p["w"].float().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros((), dtype=torch.int32, device=dev)
    pg = torch.cat([pg, torch.zeros(1, device=dev)])[:n]          # any 16-byte aligned slab start
    state = {}
    for it in range(3):
        g = rnd(n, seed=60 + it).double()
        O.adam_step(p, {"w": g}, state, 5e-4)
        ops.counter_inc(step)
        ops.adam_step(pg, g.float().to(dev), mg, vg, 5e-4, 0.5, 0.999, 1e-8, step)
    assert rel(pg, p["w"]) < 1e-6 and int(step) == 3


def test_gemm_nt_big_tile_path_with_conv_window(pkg, dev):
    """Large-tile kernel (M >= 1024, N >= 96): dilated causal conv window, strided output slice, bias + ReLU, accumulate."""
    Lm = pkg.layers
    B, T, Cc, d = 40, 34, 300, 2
    x = rnd(B, Cc, T, seed=70).double()
    w = rnd(Cc, Cc, 2, seed=71, scale=0.05).double()
    b = rnd(Cc, seed=72).double()
    y = torch.relu(F.conv1d(x, w, b, padding=d, dilation=d)[:, :, :T])
    out = Lm.conv_fwd(cl(x.float()).to(dev), Lm.pack_conv_weight(w.float().to(dev)), b.float().to(dev), 2, pad=d, dil=d, rows_out=T,
                      act_slope=0.0)
    assert rel(out, cl(y)) < 1e-5
    M, N, K = 2000, 130, 600
    xa, wa = rnd(M, K, seed=73), rnd(N, K, seed=74, scale=0.1)
    big = torch.zeros(M, 200, device=dev)
    pkg.ops.gemm_nt(pkg.ops.Win.plain(xa.to(dev)), wa.to(dev), None, big[:, 30:160])
    pkg.ops.gemm_nt(pkg.ops.Win.plain(xa.to(dev)), wa.to(dev), None, big[:, 30:160], accumulate=True)
    assert rel(big[:, 30:160], 2 * (xa.double() @ wa.double().t())) < 1e-5 and float(big[:, :30].abs().max()) == 0


@pytest.mark.parametrize("M,N,K", [(13056, 300, 600), (4352, 600, 900), (1030, 49, 68), (2049, 97, 108), (1500, 161, 1024), (1024, 48, 64)])
def test_gemm_nt_split_bf16x3_path_is_fp32_accurate(pkg, dev, M, N, K):
    """The big-product path (csrc/gemm_split.hip: fp32 operands split exactly into three bf16 terms, six partial products on the bf16
    matrix cores) against fp64 at the fp32 tolerance, every tile of its menu, ragged edges in M, N and K, bias + activation,
    accumulate, strided output -- and full-range operands (8 decades of magnitude) so that lost low-order terms would show."""
    ops, Win = pkg.ops, pkg.ops.Win
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g) * torch.pow(10.0, torch.randint(-4, 4, (M, 1), generator=g).float())
    w = torch.randn(N, K, generator=g) * 0.1
    b = torch.randn(N, generator=g)
    ref = F.leaky_relu(x.double() @ w.double().t() + b.double(), 0.3)
    out = torch.full((M, N + 7), float("nan"), device=dev)
    ops.gemm_nt(Win.plain(x.to(dev)), w.to(dev), b.to(dev), out[:, 3:3 + N], act_slope=0.3)
    row_scale = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)                 # per-row normalisation: rows span 8 decades
    assert float(((out[:, 3:3 + N].double().cpu() - ref).abs() / row_scale).max()) < 1e-5
    assert bool(torch.isnan(out[:, :3]).all()) and bool(torch.isnan(out[:, 3 + N:]).all())
    ops.gemm_nt(Win.plain(x.to(dev)), w.to(dev), None, out[:, 3:3 + N], accumulate=True)
    ref2 = ref + x.double() @ w.double().t()
    assert float(((out[:, 3:3 + N].double().cpu() - ref2).abs() / ref2.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)).max()) < 1e-5


@pytest.mark.parametrize("B,L,C,taps,step,shift,dil,rows_out,N", [(37, 60, 36, 3, 2, 1, 2, 28, 100), (5, 400, 64, 2, 1, 0, 100, 300, 49),
                                                                  (30, 40, 100, 1, 1, 3, 1, 35, 260), (1, 2200, 48, 4, 1, 0, 5, 2100, 96)])
def test_gemm_nt_split_fast_addressing_tap_windows(pkg, dev, B, L, C, taps, step, shift, dil, rows_out, N):
    """Windows WITHOUT padding take the split kernel's fast addressing (32-bit offsets, clamped rows, the tap walk as one compare-and-select per
    slab): taps narrower than a slab's reach (cw not a multiple of 32: a slab straddles two taps), row steps and shifts, several batches,
    ragged M / N / K tails, K-concatenated weight segments -- against an explicit gather in fp64."""
    ops, Win = pkg.ops, pkg.ops.Win
    g = torch.Generator().manual_seed(B * L + C + taps)
    x = torch.randn(B, L, C, generator=g)
    K = taps * C
    wbuf = torch.randn(taps, N + 3, C, generator=g) * 0.1                 # segment t = wbuf[t, :N]: seg_k = C, seg stride (N + 3) * C floats
    bias = torch.randn(N, generator=g)
    rows = torch.arange(rows_out) * step + shift
    src = rows[:, None] + torch.arange(taps)[None, :] * dil                # (rows_out, taps), all inside [0, L) by construction
    assert int(src.min()) >= 0 and int(src.max()) < L
    A = x[:, src, :].reshape(B * rows_out, K).double()                     # (B, rows_out, taps, C) -> [M][K], tap-major like the window
    Wcat = torch.cat([wbuf[t, :N] for t in range(taps)], dim=1).double()   # [N][K]
    ref = F.leaky_relu(A @ Wcat.t() + bias.double(), 0.2)
    xd, wd = x.to(dev), wbuf.to(dev)
    win = Win(xd, batches=B, batch_stride=xd.stride(0), row_stride=xd.stride(1), rows_in=L, rows_out=rows_out, cw=C, K=K, row_step=step,
              shift=shift, dil=dil)
    out = torch.full((B * rows_out, N), float("nan"), device=dev)
    ops.gemm_nt(win, wd[0, :N], bias.to(dev), out, act_slope=0.2, b_seg=(C, (N + 3) * C))
    assert rel(out, ref) < 1e-5


@pytest.mark.parametrize("M,N,K", [(4352, 300, 600), (1100, 52, 72), (1030, 49, 68)])
def test_gemm_nt_epilogue_gate_and_residual_output(pkg, dev, M, N, K):
    """Epilogue extensions of the big-product path: `gate` (keep the result where gate > 0: relu + dropout backward of the producer's saved
    activation) after act / out_scale, before accumulate; `res` + `out2` (out2 = relu(out + res): the TCN block's closing add + ReLU as a second
    output).  Vectorised and scalar (N % 4 != 0) epilogues, ragged tiles; unsupported families refuse the operands loudly."""
    ops, Win = pkg.ops, pkg.ops.Win
    g = torch.Generator().manual_seed(M + N)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1, torch.randn(N, generator=g)
    mask = (torch.rand(M, N, generator=g) > 0.3).float() / 0.7
    gate = torch.relu(torch.randn(M, N, generator=g))                      # ~half zeros, like a saved post-ReLU activation
    res = torch.randn(M, N, generator=g)
    base = torch.randn(M, N, generator=g)
    xd, wd = x.to(dev), w.to(dev)
    assert ops.nt_ext_supported(Win.plain(xd), wd, torch.empty(M, N, device=dev))
    # forward form: out = relu(x W^T + b) * mask, out2 = relu(out + res)
    out, out2 = torch.full((M, N), float("nan"), device=dev), torch.full((M, N), float("nan"), device=dev)
    ops.gemm_nt(Win.plain(xd), wd, b.to(dev), out, act_slope=0.0, out_scale=mask.to(dev), res=res.to(dev), out2=out2)
    ref = torch.relu(x.double() @ w.double().t() + b.double()) * mask.double()
    assert rel(out, ref) < 1e-5
    ref2 = torch.relu(out.double().cpu() + res.double())                    # from the kernel's own `out`: the add + ReLU must be exact
    assert float((out2.double().cpu() - ref2).abs().max()) <= 1e-6 * float(ref2.abs().max())
    # backward form: out = base + (x W^T * mask where gate > 0)
    acc = base.to(dev).clone()
    ops.gemm_nt(Win.plain(xd), wd, None, acc, out_scale=mask.to(dev), gate=gate.to(dev), accumulate=True)
    refb = base.double() + (x.double() @ w.double().t()) * mask.double() * (gate > 0).double()
    assert rel(acc, refb) < 1e-5
    zeros = (gate == 0)
    assert torch.equal(acc.cpu()[zeros], base[zeros])                        # gated-off entries contribute exactly nothing
    # a product outside the big-product family must refuse the operands instead of ignoring them
    xs, ws = torch.randn(64, 32, device=dev), torch.randn(20, 32, device=dev)
    small = torch.empty(64, 20, device=dev)
    assert not ops.nt_ext_supported(Win.plain(xs), ws, small)
    with pytest.raises(Exception):
        ops.gemm_nt(Win.plain(xs), ws, None, small, gate=torch.ones(64, 20, device=dev))


def _row_err(out, ref):
    """max over rows of |out - ref| / max|ref row| (rows of the big-shape tests span eight decades); fp64, on the reference's device."""
    o = out.double().to(ref.device)
    return float(((o - ref).abs() / ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)).max())


def _mm64(x, w):
    """x @ w^T in fp64 ON THE GPU (torch / rocBLAS dgemm: an independent fp64 reference; the 13 056-row products take seconds per call on the
    host cores and made the suite twice as long)."""
    return x.double() @ w.double().t()


def _planes_of(ops, fmt):
    """The weight-plane splitter of an operand format: 'h2' = fp16 x 2 (round 6: three matrix instructions per product), 'x3' = bf16 x 3 (six)."""
    return {"h2": ops.split2h_planes, "x3": ops.split3_planes}[fmt]


@pytest.mark.parametrize("fmt", ["h2", "x3"])
@pytest.mark.parametrize("M,N,K,tile", [(13000, 900, 600, (128, 192)), (13056, 900, 108, (128, 192)), (13056, 300, 600, (128, 160)),
                                        (9999, 596, 1000, (128, 160))])
def test_gemm_nt_mover_wave_kernel_is_fp32_accurate(pkg, dev, M, N, K, tile, fmt):
    """csrc/gemm_mw.hip (512-thread workgroups: four mover waves stage + split the operands, four matrix waves multiply; big tiles, one
    workgroup per CU) on the shapes it is chosen for -- the stacked forward's GRU input projections (multimodal_context_net.py:98-99: N =
    900, K = 600 / 108) and the TCN convs (model/tcn.py:19-46: N = 300) -- against fp64 at the fp32 tolerance: ragged tails in M, N and K,
    operands spanning 8 decades (lost low-order terms would show), bias + activation, accumulate, strided output, a two-problem group."""
    ops, Win = pkg.ops, pkg.ops.Win
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g) * torch.pow(10.0, torch.randint(-4, 4, (M, 1), generator=g).float())
    ws = [torch.randn(N, K, generator=g) * 0.1 for _ in range(2)]
    bs = [torch.randn(N, generator=g) for _ in range(2)]
    xd = x.to(dev)
    outs = [torch.full((M, N + 8), float("nan"), device=dev) for _ in range(2)]
    wd = [w.to(dev) for w in ws]
    split = _planes_of(ops, fmt)
    probs = [dict(A=Win.plain(xd), W=w, bias=b.to(dev), out=o[:, 4:4 + N], act_slope=0.3, w_planes=split(w)) for w, b, o in zip(wd, bs, outs)]
    assert ops.nt_kernel_plan([{k: v for k, v in p.items() if k != "w_planes"} for p in probs])[0] == 1      # without pre-split weights: staged-slab kernel
    plan = ops.nt_kernel_plan(probs)
    assert plan == (2,) + tile, plan                          # this test is about the mover-wave kernel: fail if the dispatcher chose another
    ops.gemm_nt_group(probs)
    for w, b, o in zip(wd, bs, outs):
        ref = F.leaky_relu(_mm64(xd, w) + b.to(dev).double(), 0.3)
        assert _row_err(o[:, 4:4 + N], ref) < 1e-5
        assert bool(torch.isnan(o[:, :4]).all()) and bool(torch.isnan(o[:, 4 + N:]).all())
    # the weights' rows may sit anywhere inside a bigger plane buffer (w_row0): both matrices stacked in one buffer, accumulate form
    both = split(torch.cat(wd, 0))
    acc_p = [dict(A=Win.plain(xd), W=wd[1], bias=None, out=outs[0][:, 4:4 + N], accumulate=True, w_planes=both, w_row0=N)]
    assert ops.nt_kernel_plan(acc_p)[0] == 2
    ops.gemm_nt_group(acc_p)
    ref2 = F.leaky_relu(_mm64(xd, wd[0]) + bs[0].to(dev).double(), 0.3) + _mm64(xd, wd[1])
    assert _row_err(outs[0][:, 4:4 + N], ref2) < 1e-5


@pytest.mark.parametrize("fmt", ["h2", "x3"])
def test_gemm_nt_mover_wave_kernel_windows_and_epilogues(pkg, dev, fmt):
    """The same kernel behind the windows and epilogues of the text encoder at the stacked forward's size (B_s = 384 clips x 34 frames):
    the dilated causal conv with its zero padding (rows before the clip read as zero through the buffer descriptor's range check),
    ReLU + dropout scale + the block's closing relu(out + x) as second output (forward form, model/tcn.py:27-46), the input-gradient
    form with reversed taps (negative dilation), gate and accumulate, and K-concatenated weight segments over a two-tap window."""
    ops, Win, Lm = pkg.ops, pkg.ops.Win, pkg.layers
    B, T, Cc, d = 384, 34, 300, 4
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cc, T, generator=g).double()
    w = (torch.randn(Cc, Cc, 2, generator=g) * 0.05).double()
    b = torch.randn(Cc, generator=g).double()
    mask = ((torch.rand(B, T, Cc, generator=g) > 0.3).float() / 0.7)
    res = torch.randn(B, T, Cc, generator=g)
    xd = cl(x.float()).to(dev)
    wp = Lm.pack_conv_weight(w.float().to(dev))
    a_win = Win.conv(xd, 2, pad=d, dil=d, rows_out=T)
    out, out2 = torch.full((B, T, Cc), float("nan"), device=dev), torch.full((B, T, Cc), float("nan"), device=dev)
    split = _planes_of(ops, fmt)
    wpl = split(wp)
    assert ops.nt_kernel_plan([dict(A=a_win, W=wp, bias=b.float().to(dev), out=out, c_batch_stride=out.stride(0), c_row_stride=out.stride(1),
                                    c_rows_out=T, w_planes=wpl)])[0] == 2
    o = Lm.conv_fwd(xd, wp, b.float().to(dev), 2, pad=d, dil=d, rows_out=T, act_slope=0.0, out_scale=mask.to(dev), res=res.to(dev), out2=out2, out=out,
                    w_planes=wpl)
    y = torch.relu(F.conv1d(x, w, b, padding=d, dilation=d)[:, :, :T])
    ref = cl(y) * mask.double()
    assert rel(o, ref) < 1e-5
    ref2 = torch.relu(o.double().cpu() + res.double())
    assert float((out2.double().cpu() - ref2).abs().max()) <= 1e-6 * float(ref2.abs().max())
    # input-gradient form: dx[t] = dy[t] . W[:, :, 1] + dy[t + d] . W[:, :, 0] (rows past the clip read as zero), gated, accumulated
    dy = torch.randn(B, T, Cc, generator=g)
    gate = torch.relu(torch.randn(B * T, Cc, generator=g))
    base = torch.randn(B * T, Cc, generator=g)
    wT = torch.cat([w[:, :, 1].t().contiguous(), w[:, :, 0].t().contiguous()], dim=1).float()      # [Ci][2 Co]: taps (t, t + d)
    dyd = dy.to(dev)
    a_back = Win.taps(dyd, 2, shift=0, dil=d, rows_out=T)
    acc = base.to(dev).clone()
    wTd = wT.to(dev)
    probs = [dict(A=a_back, W=wTd, bias=None, out=acc, gate=gate.to(dev), accumulate=True, w_planes=split(wTd))]
    assert ops.nt_kernel_plan(probs)[0] == 2
    ops.gemm_nt_group(probs)
    dyp = torch.cat([dy.double(), torch.zeros(B, d, Cc, dtype=torch.float64)], dim=1)
    full = dyp[:, :T] @ w[:, :, 1] + dyp[:, d:d + T] @ w[:, :, 0]                                   # (B, T, Ci)
    refb = base.double() + full.reshape(B * T, Cc) * (gate > 0).double()
    assert rel(acc, refb) < 1e-5
    assert torch.equal(acc.cpu()[gate == 0], base[gate == 0])
    # K-concatenated weights over a two-"tap" window: dx = dgi_fwd @ W_fwd + dgi_rev @ W_rev at the stacked size
    Mh, Kh, N = 13056, 448, 600
    dgi = (torch.randn(2, Mh, Kh, generator=g)).to(dev)
    wbuf = (torch.randn(3, N, Kh, generator=g) * 0.1).to(dev)
    a_cat = Win(dgi, batches=1, batch_stride=0, row_stride=Kh, rows_in=2 * Mh, rows_out=Mh, cw=Kh, K=2 * Kh, dil=Mh)
    outc = torch.full((Mh, N), float("nan"), device=dev)
    assert ops.nt_kernel_plan([dict(A=a_cat, W=wbuf[0], bias=None, out=outc, b_seg=(Kh, 2 * N * Kh))])[0] == 1        # weight segments: staged-slab kernel
    ops.gemm_nt(a_cat, wbuf[0], None, outc, b_seg=(Kh, 2 * N * Kh))
    refc = _mm64(dgi[0], wbuf[0]) + _mm64(dgi[1], wbuf[2])
    assert float((outc.double() - refc).abs().max() / refc.abs().max()) < 1e-5


@pytest.mark.parametrize("fmt", ["h2", "x3"])
def test_regenerated_dropout_equals_stored_mask(pkg, dev, fmt):
    """ops.Drop: the text encoder's dropout scale masks are not stored -- the conv epilogues (mover-wave and staged-slab kernels), the gated
    input-gradient epilogue and act_mask_bwd / act_mask_bwd2 regenerate their elements from the counter RNG (model/tcn.py:22-29 forward and
    backward).  Every consumer must produce BIT-IDENTICAL results to the stored mask of the same draw (tg_dropout_mask), including on the row
    slice the backward works on and at a non-zero index offset (conv j of the eight shares one site)."""
    ops, Win, Lm = pkg.ops, pkg.ops.Win, pkg.layers
    g = torch.Generator().manual_seed(21)
    B, T, Cc, d = 384, 34, 300, 2
    state = ops.new_rng_state(1234, dev)
    drop = ops.Drop(state, 7, 0.3, (B, T, Cc), index0=2 * B * T * Cc)
    mask = drop.materialize()
    assert mask.shape == (B, T, Cc) and 0.25 < float((mask == 0).float().mean()) < 0.35 and float(mask.max()) == pytest.approx(1 / 0.7)
    x = torch.randn(B, T, Cc, generator=g).to(dev)
    wp = (torch.randn(Cc, 2 * Cc, generator=g) * 0.05).to(dev)
    b = torch.randn(Cc, generator=g).to(dev)
    wpl = _planes_of(ops, fmt)(wp)
    res = torch.randn(B, T, Cc, generator=g).to(dev)
    outs = []
    for m in (mask, drop):                                  # forward conv on the mover-wave kernel, with the residual second output
        o, o2 = torch.empty(B, T, Cc, device=dev), torch.empty(B, T, Cc, device=dev)
        probs = [dict(A=Win.conv(x, 2, pad=d, dil=d, rows_out=T), W=wp, bias=b, out=o, act_slope=0.0, out_scale=m, res=res, out2=o2, res_slope=0.0,
                      c_batch_stride=o.stride(0), c_row_stride=o.stride(1), c_rows_out=T, w_planes=wpl)]
        assert ops.nt_kernel_plan(probs)[0] == 2
        ops.gemm_nt_group(probs)
        outs.append((o, o2))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float((outs[0][0] == 0).float().mean()) > 0.25
    # backward forms on the rows of the differentiated call
    rows = slice(128, 256)
    M = 128 * T
    dy, y, o1 = (torch.randn(M, Cc, generator=g).to(dev) for _ in range(3))
    wT = (torch.randn(Cc, 2 * Cc, generator=g) * 0.05).to(dev)
    dc3 = torch.randn(128, T, Cc, generator=g).to(dev)
    gate = torch.relu(torch.randn(M, Cc, generator=g)).to(dev)
    got = []
    for m in (mask, drop):
        mr = m[rows].reshape(M, -1)
        dsum, dc = ops.act_mask_bwd2(dy, y, o1, mr, 0.0, torch.empty_like(dy), torch.empty_like(dy))
        dx = ops.act_mask_bwd(dy, y, mr, 0.0, torch.empty_like(dy))
        dh = torch.empty(M, Cc, device=dev)
        pr = [dict(A=Win.taps(dc3, 2, shift=d, dil=-d, rows_out=T), W=wT, bias=None, out=dh, out_scale=mr, gate=gate)]
        assert ops.nt_kernel_plan(pr)[0] == 1               # 4 352 rows: staged-slab kernel
        ops.gemm_nt_group(pr)
        got.append((dsum, dc, dx, dh))
    for a, b_ in zip(*got):
        assert torch.equal(a, b_)
    # draw-and-apply without storing the mask (embedding dropout, the generator GRU's inter-layer dropout): same y, same mask when regenerated,
    # and x * mask through the regenerating multiply
    xe = torch.randn(B, T, Cc, generator=g).to(dev)
    y1, m1 = ops.dropout_apply(xe, 0.1, state, 9)
    y2, d2 = ops.dropout_apply(xe, 0.1, state, 9, store_mask=False)
    assert isinstance(d2, ops.Drop) and torch.equal(y1, y2) and torch.equal(d2.materialize(), m1)
    dyr = torch.randn(M, Cc, generator=g).to(dev)
    assert torch.equal(ops.mul(dyr, d2[rows].reshape(M, -1).contiguous(), torch.empty_like(dyr)), dyr * m1[rows].reshape(M, -1))
    # embedding look-up + dropout in one pass == look-up, then the stored-mask dropout of the same site
    table = torch.randn(2000, Cc, generator=g).to(dev)
    idx = torch.randint(-1, 2000, (B * T,), generator=g).to(dev)
    ye, de = ops.embed_gather_drop(table, idx, torch.empty(B, T, Cc, device=dev), 0.1, state, 9)
    emb = ops.embed_gather(table, idx, torch.empty(B, T, Cc, device=dev))
    assert torch.equal(ye, emb * m1) and torch.equal(de.materialize(), m1)
    # refused where no kernel regenerates it (a small product), loudly
    xs, ws, os_ = torch.randn(64, 64, device=dev), torch.randn(32, 64, device=dev), torch.empty(64, 32, device=dev)
    with pytest.raises(Exception):
        ops.gemm_nt(Win.plain(xs), ws, None, os_, out_scale=ops.Drop(state, 7, 0.3, (64, 32)))


def test_iter_head_equals_the_separate_launches(pkg, dev):
    """tg_iter_head (the head of train_iter_gan, train_gan.py:13-30,50,67-72, one launch) against the seven launches it replaces: counters,
    stacked seed poses, stacked word ids, stacked speaker ids with the last copy shuffled by the permutation drawn at the NEW rng step."""
    ops = pkg.ops
    B, T, D, ng = 128, 34, 27, 3
    g = torch.Generator().manual_seed(3)
    target = torch.randn(B, T, D, generator=g).to(dev)
    text = torch.randint(0, 20000, (B, T), generator=g).to(dev)
    vid = torch.randint(0, 1370, (B,), generator=g).to(dev)
    for permute, injected in ((True, False), (True, True), (False, False)):
        ra, rb = ops.new_rng_state(11, dev), ops.new_rng_state(12, dev)
        ra2, rb2 = ra.clone(), rb.clone()
        ca, cb = torch.zeros((), dtype=torch.int32, device=dev), torch.zeros((), dtype=torch.int32, device=dev)
        ca2, cb2 = ca.clone(), cb.clone()
        perm_in = torch.randperm(B, generator=g).to(dev) if injected else None
        pre_s, text_s, vid_s = ops.iter_head(ra, rb, ca, cb, target, 4, ng, text=text, vid=vid, permute_last=permute, perm_in=perm_in, perm_site=5)
        ops.iter_begin(ra2, rb2, ca2, cb2)
        pre = ops.make_pre_seq(target, torch.empty(B, T, D + 1, device=dev), 4)
        perm = perm_in if injected else ops.randperm(torch.empty(B, dtype=torch.int64, device=dev), ra2, 5)
        last = ops.gather_i64(vid, perm, torch.empty_like(vid)) if permute else vid
        assert torch.equal(ra, ra2) and torch.equal(rb, rb2) and int(ca) == int(ca2) == 1 and int(cb) == int(cb2) == 1
        assert torch.equal(pre_s, pre.repeat(ng, 1, 1)) and torch.equal(text_s, text.repeat(ng, 1))
        assert torch.equal(vid_s, torch.cat([vid] * (ng - 1) + [last]))
        if permute and not injected:
            assert sorted(perm.tolist()) == list(range(B)) and perm.tolist() != list(range(B))
    # seed poses written straight into the pose columns of wider rows (the GRU input buffer)
    ra = ops.new_rng_state(11, dev)
    pre_w, _, _ = ops.iter_head(ra, None, None, None, target, 4, ng, text=text, row_floats=108)
    assert pre_w.stride() == (T * 108, 108, 1) and torch.equal(pre_w, ops.make_pre_seq(target, torch.empty(B, T, D + 1, device=dev), 4).repeat(ng, 1, 1))
    # no speaker ids, one copy, only one counter
    ra = ops.new_rng_state(11, dev)
    pre_s, text_s, vid_s = ops.iter_head(ra, None, None, None, target, 4, 1, text=text)
    assert vid_s is None and int(ra[1]) == 1 and torch.equal(pre_s, ops.make_pre_seq(target, torch.empty(B, T, D + 1, device=dev), 4))
    # the real half of the discriminator's stacked input, copied by the same launch (torch.cat of train_gan.py:30-31)
    d_in = torch.full((3 * B, T, D), -7.0, device=dev)
    ops.iter_head(ops.new_rng_state(11, dev), None, None, None, target, 4, 2, text=text, target_copy=d_in[:B])
    assert torch.equal(d_in[:B], target) and bool((d_in[B:] == -7.0).all())


@pytest.mark.parametrize("M,K", [(7168, 192), (3584, 192), (37, 64)])
def test_narrow8_pair_matches_fp64(pkg, dev, M, K):
    """tg_narrow8_pair: dx = dgi_fwd W_ih_fwd + dgi_rev W_ih_rev for a GRU layer with 8 input channels (the discriminator's nn.GRU(8, 64),
    multimodal_context_net.py:222-223 backward), W_ih as stored ([3H][8])."""
    ops = pkg.ops
    a0, a1 = rnd(M, K, seed=1).to(dev), rnd(M, K, seed=2).to(dev)
    w0, w1 = rnd(K, 8, seed=3, scale=0.2).to(dev), rnd(K, 8, seed=4, scale=0.2).to(dev)
    out = ops.narrow8_pair(a0, a1, w0, w1, torch.full((M, 8), float("nan"), device=dev))
    ref = a0.double() @ w0.double() + a1.double() @ w1.double()
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 1e-5


def test_bf16_math_mode_tier(pkg, dev):
    """tg_set_math_mode(1): the big forward / input-gradient products take bf16 operands (one MFMA per product, fp32 accumulate).
    Op-level error at the bf16 level (and clearly different from the fp32 result: the mode really switches), and one full GAN
    iteration at B = 32 against the fp64 oracle within the bf16 tolerances of SURVEY Q14 (2e-2 forward / 5e-2 gradients)."""
    ops, Win = pkg.ops, pkg.ops.Win
    M, N, K = 4352, 300, 600
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1)
    ref = x.double() @ w.double().t()
    out32, out16 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.gemm_nt(Win.plain(x.to(dev)), w.to(dev), None, out32)
    assert ops.get_math_mode() == "f32"
    try:
        ops.set_math_mode("bf16")
        assert ops.get_math_mode() == "bf16"
        ops.gemm_nt(Win.plain(x.to(dev)), w.to(dev), None, out16)
        from tests.harness import run_train_parity
        # B = 32: the stacked forward (3 x 32 x 34 rows) and the backward (32 x 34 rows) are big enough for the bf16 path (M >= 1024)
        worst = run_train_parity(pkg, dev, batch=32, epochs=(11,), seed=77, check_step=False)
    finally:
        ops.set_math_mode("f32")
    e32, e16 = rel(out32, ref), rel(out16, ref)
    assert e32 < 1e-5 and 1e-4 < e16 < 2e-2, (e32, e16)
    assert worst < 5e-2, worst
    with pytest.raises(Exception):
        ops.set_math_mode(2)


@pytest.mark.parametrize("B,T,rows", [(5, 28, 8), (70, 28, 8), (256, 28, 8), (70, 28, 16), (256, 28, 16), (33, 7, 8), (33, 3, 16), (33, 3, 8), (17, 2, 8),
                                      (17, 2, 16), (9, 1, 8), (1100, 5, 0)])
def test_gru_h64_stack_with_fused_dropout(pkg, dev, B, T, rows, monkeypatch):
    """The discriminator's GRU stack (4 layers, H = 64) with injected inter-layer dropout masks against a layer-by-layer
    nn.GRU fp64 reference that multiplies the same masks in between: forward output, input gradient, every weight gradient.  The
    masks ride inside the recurrence kernels (y_drop = y * mask in the forward, dy * mask in the backward load).  Both workgroup sizes of the
    kernels (8 and 16 batch rows, TG_H64_ROWS; 0 = the library's choice: 16 above 1 024 rows), odd and very short sequences (down to one
    step), ragged batch tiles."""
    if rows:
        monkeypatch.setenv("TG_H64_ROWS", str(rows))
    else:
        monkeypatch.delenv("TG_H64_ROWS", raising=False)
    Lm = pkg.layers
    H, L = 64, 4
    layers = [torch.nn.GRU(8 if l == 0 else 2 * H, H, num_layers=1, batch_first=True, bidirectional=True).double() for l in range(L)]
    g = torch.Generator().manual_seed(B)
    masks = [(torch.rand(B, T, 2 * H, generator=g) >= 0.3).double() / 0.7 for _ in range(L - 1)]
    x = rnd(B, T, 8, seed=31).double().requires_grad_(True)
    cur = x
    for l in range(L):
        cur, _ = layers[l](cur)
        if l < L - 1:
            cur = cur * masks[l]
    dy = rnd(B, T, 2 * H, seed=32).double()
    cur.backward(dy)
    P = {}
    for l in range(L):
        for k, v in layers[l].named_parameters():
            P["gru." + k.replace("_l0", f"_l{l}")] = v.detach().float().to(dev).contiguous()
    G = {k: torch.zeros_like(v) for k, v in P.items()}
    inject = {f"d.gru.drop{l}": masks[l].float().to(dev) for l in range(L - 1)}
    yg, tape = Lm.gru_stack_fwd(x.detach().float().to(dev), P, "gru", L, H, p_drop=0.3, training=True, save=True, inject=inject, tag="d")
    assert rel(yg, cur) < 1e-5
    dx = Lm.gru_stack_bwd(dy.float().to(dev), tape, P, G, "gru", L)
    assert rel(dx, x.grad) < 1e-4
    for l in range(L):
        for k, v in layers[l].named_parameters():
            assert rel(G["gru." + k.replace("_l0", f"_l{l}")], v.grad) < 1e-4, (l, k)
    # drawn masks (no injection): keep-rate and scale, and y_drop == y * mask
    class R:
        def __init__(self): self.state = pkg.ops.new_rng_state(11, dev); self.n = {}
        def site(self, name): return self.n.setdefault(name, len(self.n) + 1)
    y2, tape2 = Lm.gru_stack_fwd(x.detach().float().to(dev), P, "gru", L, H, p_drop=0.3, training=True, save=True, rng=R(), tag="d")
    for l in range(L - 1):
        m = tape2.masks[l]
        vals = torch.unique(m)
        assert vals.numel() == 2 and float(vals[0]) == 0 and abs(float(vals[1]) - 1 / 0.7) < 1e-6
        if B >= 70:
            assert abs(float((m > 0).float().mean()) - 0.7) < 0.02
        assert torch.equal(tape2.x[l + 1], tape2.y[l] * m)


def test_grouped_gemm_launches_and_concatenated_k(pkg, dev):
    """tg_gemm_nt_group / tg_gemm_tn_group: several products in one launch == the same products one by one == fp64; problems of
    different kernel families in one call are partitioned by the wrapper; K-concatenated weights (b_seg) with a two-tap A window
    reproduce dx = dgi_fwd @ W_fwd + dgi_rev @ W_rev of the GRU backward."""
    ops, Win = pkg.ops, pkg.ops.Win
    M, K = 2100, 108
    x = rnd(M, K, seed=1).to(dev)
    ws = [rnd(n, K, seed=10 + i, scale=0.1).to(dev) for i, n in enumerate((900, 900, 20, 70))]      # big, big, narrow, big
    bs = [rnd(w.shape[0], seed=20 + i).to(dev) for i, w in enumerate(ws)]
    outs = [torch.full((M, w.shape[0]), float("nan"), device=dev) for w in ws]
    ops.gemm_nt_group([dict(A=Win.plain(x), W=w, bias=b, out=o, act_slope=0.3) for w, b, o in zip(ws, bs, outs)])
    for w, b, o in zip(ws, bs, outs):
        ref = F.leaky_relu(x.double().cpu() @ w.double().cpu().t() + b.double().cpu(), 0.3)
        assert rel(o, ref) < 1e-5
    # K-concatenation: A = [2][M][Kh] seen as two taps, weights = two [N][Kh] segments somewhere in memory
    Mh, Kh, N = 1500, 192, 128
    dgi = rnd(2, Mh, Kh, seed=30).to(dev)
    wbuf = rnd(3, N, Kh, seed=31, scale=0.1).to(dev)                      # segment 0 = wbuf[0], segment 1 = wbuf[2] (stride 2*N*Kh floats)
    a_cat = Win(dgi, batches=1, batch_stride=0, row_stride=Kh, rows_in=2 * Mh, rows_out=Mh, cw=Kh, K=2 * Kh, dil=Mh)
    out = torch.full((Mh, N), float("nan"), device=dev)
    ops.gemm_nt(a_cat, wbuf[0], None, out, b_seg=(Kh, 2 * N * Kh))
    ref = dgi[0].double().cpu() @ wbuf[0].double().cpu().t() + dgi[1].double().cpu() @ wbuf[2].double().cpu().t()
    assert rel(out, ref) < 1e-5
    # four weight gradients of different shapes in one launch, with bias gradients
    Mt = 1088
    dys = [rnd(Mt, n, seed=40 + i).to(dev) for i, n in enumerate((192, 192, 96, 33))]
    xs = [rnd(Mt, k, seed=50 + i).to(dev) for i, k in enumerate((128, 64, 200, 7))]
    dws = [torch.zeros(dy.shape[1], xx.shape[1], device=dev) for dy, xx in zip(dys, xs)]
    dbs = [torch.zeros(dy.shape[1], device=dev) for dy in dys]
    ops.gemm_tn_group([dict(dY=dy, A=Win.plain(xx), dW=dw, dbias=db) for dy, xx, dw, db in zip(dys, xs, dws, dbs)])
    for dy, xx, dw, db in zip(dys, xs, dws, dbs):
        assert rel(dw, dy.double().cpu().t() @ xx.double().cpu()) < 1e-5 and rel(db, dy.double().cpu().sum(0)) < 1e-5


@pytest.mark.parametrize("M,N,K", [(4352, 900, 600), (1088, 52, 76), (2176, 300, 108), (1500, 64, 48)])
def test_gemm_tn_split_bf16x3_path_is_fp32_accurate(pkg, dev, M, N, K):
    """Weight gradients on the bf16 matrix cores (three-way split operands, hardware-transposed LDS reads, csrc/gemm_split.hip) against
    fp64 at the fp32 tolerance: ragged N / K (multiples of 4, not of the 64-wide tile), rows spanning 8 decades, bias gradient, and a
    second call accumulating on top of the first."""
    ops, Win = pkg.ops, pkg.ops.Win
    g = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(M, N, generator=g) * torch.pow(10.0, torch.randint(-4, 3, (1, N), generator=g).float())
    x = torch.randn(M, K, generator=g)
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    ops.gemm_tn(dy.to(dev), Win.plain(x.to(dev)), dw, dbias=db)
    ref = dy.double().t() @ x.double()
    scale = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)              # per output row: the dY columns span 7 decades
    assert float(((dw.double().cpu() - ref).abs() / scale).max()) < 1e-5
    assert rel(db, dy.double().sum(0)) < 1e-5
    ops.gemm_tn(dy.to(dev), Win.plain(x.to(dev)), dw, dbias=db)
    assert float(((dw.double().cpu() - 2 * ref).abs() / scale).max()) < 1e-5


@pytest.fixture(params=["h2", "x3"])
def operand_format(request, pkg):
    """Both operand formats of the mover-wave products: 'h2' = fp16 x 2 (the default: three matrix instructions per product, per-row /
    per-column power-of-two scales), 'x3' = bf16 x 3 (six; TG_GEMM_H2=0)."""
    prev = pkg.ops.GEMM_H2, pkg.ops.TN_AUTO_COLMAX
    pkg.ops.GEMM_H2 = pkg.ops.TN_AUTO_COLMAX = request.param == "h2"          # (weight gradients: column magnitudes measured by a pass of their own)
    yield request.param
    pkg.ops.GEMM_H2, pkg.ops.TN_AUTO_COLMAX = prev


def test_gemm_tn_mover_wave_kernel(pkg, dev, operand_format):
    """csrc/gemm_tn_mw.hip on the groups it is chosen for: the four weight gradients of a GRU layer at B = 128 (dW_ih, dW_hh of both
    directions with their bias gradients -- the bias gradient rides in the product as a column of ones; multimodal_context_net.py:98-99
    backward) and a text-encoder-sized group over a two-tap conv window with padding (model/tcn.py:19-46 backward), against fp64:
    accumulation into non-zero dW / dbias, ragged M (row splits), N and K tails, 8-decade operands."""
    ops, Win = pkg.ops, pkg.ops.Win
    g = torch.Generator().manual_seed(21)
    M, H = 4352 - 37, 300
    dgi = [(torch.randn(M, 3 * H, generator=g) * torch.pow(10.0, torch.randint(-3, 3, (M, 1), generator=g).float())).to(dev) for _ in range(2)]
    x = torch.randn(M, 2 * H, generator=g).to(dev)
    hp = torch.randn(M, H, generator=g).to(dev)
    # both combines of the row splits: workspace + fixed-order second pass (ops.TN_MW_WS, the default: two runs must agree bit for bit) and
    # float atomics
    prev = ops.TN_MW_WS
    try:
        for use_ws in (True, True, False):
            ops.TN_MW_WS = use_ws
            gq = torch.Generator().manual_seed(22)
            probs, refs = [], []
            for d in range(2):
                for A, Kc in ((x, 2 * H), (hp, H)):
                    dW = torch.randn(3 * H, Kc, generator=gq).to(dev)
                    db = torch.randn(3 * H, generator=gq).to(dev)
                    refs.append((dW.double() + dgi[d].double().t() @ A.double(), db.double() + dgi[d].double().sum(0)))
                    probs.append(dict(dY=dgi[d], A=Win.plain(A), dW=dW, dbias=db))
            assert ops.tn_kernel_plan(probs) == 2
            ops.gemm_tn_group(probs)
            for p, (rw, rb) in zip(probs, refs):
                e_w = float((p["dW"].double() - rw).abs().max() / rw.abs().max())
                e_b = float((p["dbias"].double() - rb).abs().max() / rb.abs().max())
                assert e_w < 1e-5 and e_b < 1e-5, (use_ws, e_w, e_b)
            if use_ws:
                got = [(p["dW"].clone(), p["dbias"].clone()) for p in probs]
                if "first" in locals():
                    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(first, got)), "workspace combine is not reproducible"
                first = got
    finally:
        ops.TN_MW_WS = prev
    # conv window with causal padding: dW[co][tap * Ci + ci] += sum_(b, t) dy[b][t][co] * x[b][t - (1 - tap) * d][ci]  (zero before the clip)
    B, T, C, d = 128, 34, 300, 4
    xs = torch.randn(B, T, C, generator=g).to(dev)
    cprobs, crefs = [], []
    for j in range(3):
        dy = torch.randn(B * T, C, generator=g).to(dev)
        dW = torch.zeros(C, 2 * C, device=dev)
        db = torch.zeros(C, device=dev)
        xpad = torch.cat([torch.zeros(B, d, C, device=dev, dtype=torch.float64), xs.double()], dim=1)            # row t + d = x[t]
        a_cat = torch.cat([xpad[:, :T], xpad[:, d:d + T]], dim=2).reshape(B * T, 2 * C)                             # taps (t - d, t)
        crefs.append((dy.double().t() @ a_cat, dy.double().sum(0)))
        cprobs.append(dict(dY=dy, A=Win.conv(xs, 2, pad=d, dil=d, rows_out=T), dW=dW, dbias=db))
    assert ops.tn_kernel_plan(cprobs) == 2
    ops.gemm_tn_group(cprobs)
    for p, (rw, rb) in zip(cprobs, crefs):
        assert float((p["dW"].double() - rw).abs().max() / rw.abs().max()) < 1e-5
        assert float((p["dbias"].double() - rb).abs().max() / rb.abs().max()) < 1e-5
    # small groups stay on the staged-slab kernel
    assert ops.tn_kernel_plan([dict(dY=dgi[0][:1100, :52], A=Win.plain(x[:1100, :76]), dW=torch.zeros(52, 76, device=dev))]) in (0, 1)
    # tg_set_tn_workgroup_cap (layers.gru_stack_bwd side_split): the last 61 clips' rows of a GRU layer's group planned for the 96 CUs the
    # cluster recurrence leaves free -- same kernel, same answers, the cap gone afterwards
    rows = slice(M - 61 * 34, M)
    gq = torch.Generator().manual_seed(23)
    probs, refs = [], []
    for d in range(2):
        for A, Kc in ((x, 2 * H), (hp, H)):
            dW, db = torch.randn(3 * H, Kc, generator=gq).to(dev), torch.randn(3 * H, generator=gq).to(dev)
            refs.append((dW.double() + dgi[d][rows].double().t() @ A[rows].double(), db.double() + dgi[d][rows].double().sum(0)))
            probs.append(dict(dY=dgi[d][rows], A=Win.plain(A[rows]), dW=dW, dbias=db))
    with ops.tn_workgroup_cap(96):
        assert pkg._lib.load().tg_get_tn_workgroup_cap() == 96 and ops.tn_kernel_plan(probs) == 2
        ops.gemm_tn_group(probs)
    assert pkg._lib.load().tg_get_tn_workgroup_cap() == 0
    for p, (rw, rb) in zip(probs, refs):
        assert float((p["dW"].double() - rw).abs().max() / rw.abs().max()) < 1e-5
        assert float((p["dbias"].double() - rb).abs().max() / rb.abs().max()) < 1e-5


def test_gemm_tn_split_with_conv_window(pkg, dev):
    """The same kernel behind a dilated causal conv window (the TCN weight gradient, M = B*T = 1360 rows) and the shifted h_{t-1} view of
    the GRU's W_hh gradient."""
    Lm, ops, Win = pkg.layers, pkg.ops, pkg.ops.Win
    B, T, Cc, d = 40, 34, 300, 4
    x = rnd(B, Cc, T, seed=80).double().requires_grad_(True)
    w = rnd(Cc, Cc, 2, seed=81, scale=0.05).double().requires_grad_(True)
    y = F.conv1d(x, w, None, padding=d, dilation=d)[:, :, :T]
    dyt = rnd(B, Cc, T, seed=82).double()
    y.backward(dyt)
    dwp = torch.zeros(Cc, 2 * Cc, device=dev)
    dc = cl(dyt.float()).to(dev).reshape(B * T, Cc)
    ops.gemm_tn(dc, Win.conv(cl(x.detach().float()).to(dev), 2, pad=d, dil=d, rows_out=T), dwp)
    want = w.grad.permute(0, 2, 1).reshape(Cc, 2 * Cc)                       # packed [Co][tap * Ci + ci]
    assert rel(dwp, want) < 1e-5
    H = 300
    yl = rnd(B, T, 2 * H, seed=83).to(dev)
    gh = rnd(B * T, 3 * H, seed=84).to(dev)
    dwh = torch.zeros(3 * H, H, device=dev)
    hwin = Win.taps(yl[:, :, :H], 1, shift=-1, dil=1, rows_out=T)
    ops.gemm_tn(gh, hwin, dwh)
    hprev = torch.cat([torch.zeros(B, 1, H, device=dev), yl[:, :-1, :H]], dim=1).reshape(B * T, H)
    assert rel(dwh, gh.double().cpu().t() @ hprev.double().cpu()) < 1e-5


# ------------------------------------------------------------------------------------------------ WavEncoder front end
@pytest.mark.parametrize("B,L,stride,pad,groups", [(3, 333, 5, 40, 1), (4, 1207, 5, 160, 2), (2, 36267, 5, 1600, 1)])
def test_wav_front_conv_bn_lrelu_fused(pkg, dev, B, L, stride, pad, groups):
    """Conv1d(1,16,15,stride,pad) -> BatchNorm1d(16) -> LeakyReLU(0.3) (multimodal_context_net.py:13-15) recomputed from the raw audio
    (csrc/audio.hip) against torch fp64 autograd: output, running statistics, and every gradient of the block; frame counts that are
    not a multiple of the 16-frame tile, statistics groups, eval mode."""
    Lm = pkg.layers
    audio = rnd(B, L, seed=31)
    audio[:, : L // 7] *= 0.05                     # a quiet stretch: uneven statistics along time
    w, b = rnd(16, 1, 15, seed=32, scale=0.3), rnd(16, seed=33, scale=0.2)
    ga, be = 1.0 + rnd(16, seed=34, scale=0.2), rnd(16, seed=35, scale=0.3)
    rm0, rv0 = rnd(16, seed=36, scale=0.1), 1.0 + rnd(16, seed=37, scale=0.1).abs()
    per = B // groups
    # reference, one statistics group after the other (stacked forward calls of the same module)
    P = [t.double().requires_grad_(True) for t in (w, b, ga, be)]
    rm, rv = rm0.double().clone(), rv0.double().clone()
    ys = []
    for g in range(groups):
        c = F.conv1d(audio[g * per:(g + 1) * per].double().unsqueeze(1), P[0], P[1], stride=stride, padding=pad)
        ys.append(F.leaky_relu(F.batch_norm(c, rm, rv, P[2], P[3], training=True, momentum=0.1, eps=1e-5), 0.3))
    y_ref = torch.cat(ys, 0)                                            # (B, 16, T1)
    T1 = y_ref.shape[2]
    dy = rnd(B, 16, T1, seed=38)
    d = lambda t: t.to(dev).contiguous()
    rm_d, rv_d, nbt = d(rm0), d(rv0), torch.zeros((), dtype=torch.int64, device=dev)
    wd, bd, gad, bed = d(w), d(b), d(ga), d(be)
    y, st = Lm.wav_front_fwd(d(audio), wd, bd, gad, bed, rm_d, rv_d, nbt, stride=stride, pad=pad, training=True, groups=groups)
    assert tuple(y.shape) == (B, T1, 16) and int(nbt) == groups
    assert rel(y, cl(y_ref)) < 1e-5
    assert rel(rm_d, rm) < 1e-5 and rel(rv_d, rv) < 1e-5
    # backward, group by group (the engine differentiates one group of a stacked forward)
    for g in range(groups):
        grads = torch.autograd.grad(ys[g], P, dy[g * per:(g + 1) * per].double(), retain_graph=True)
        dW, db, dga, dbe = (torch.zeros(16, 1, 15, device=dev), torch.zeros(16, device=dev), torch.zeros(16, device=dev), torch.zeros(16, device=dev))
        Lm.wav_front_bwd(d(cl(dy[g * per:(g + 1) * per])), st, d(audio), wd, bd, gad, dW, db, dga, dbe, g0=g, row0=g * per)
        assert rel(dW, grads[0]) < 1e-4 and rel(dga, grads[2]) < 1e-4 and rel(dbe, grads[3]) < 1e-4, (rel(dW, grads[0]), rel(dga, grads[2]), rel(dbe, grads[3]))
        assert float(db.abs().max()) < 1e-4 * float(grads[0].abs().max())          # zero by construction (bias in front of a train-mode BatchNorm)
        Lm.wav_front_bwd(d(cl(dy[g * per:(g + 1) * per])), st, d(audio), wd, bd, gad, dW, None, None, dbe, g0=g, row0=g * per)   # accumulates; NULL outputs
        assert rel(dW, 2 * grads[0]) < 1e-4 and rel(dbe, 2 * grads[3]) < 1e-4
    # second form: conv2's input gradient formed inside the reduction (feat_extractor[3] = Conv1d(16, 32, 15, stride 6))
    if T1 >= 15:
        w2 = rnd(32, 16, 15, seed=39, scale=0.1)
        for g in range(groups):
            c2 = F.conv1d(ys[g], w2.double(), None, stride=6)
            dc2 = rnd(per, 32, c2.shape[2], seed=40 + g)
            grads = torch.autograd.grad(c2, P, dc2.double(), retain_graph=True)
            dW, db, dga, dbe = (torch.zeros(16, 1, 15, device=dev), torch.zeros(16, device=dev), torch.zeros(16, device=dev), torch.zeros(16, device=dev))
            Lm.wav_front_bwd_fused(d(cl(dc2)), d(w2), st, d(audio), wd, bd, gad, dW, db, dga, dbe, g0=g, row0=g * per)
            errs = (rel(dW, grads[0]), rel(dga, grads[2]), rel(dbe, grads[3]))
            assert max(errs) < 1e-4, errs
            assert float(db.abs().max()) < 1e-4 * float(grads[0].abs().max())
    # the generic launches (window GEMM + BatchNorm kernels) agree with the fused block to rounding
    c2 = Lm.conv_fwd(d(audio)[:per].unsqueeze(2), wd.view(16, 15), bd, 15, stride=stride, pad=pad)
    y2, _ = Lm.bn_fwd(c2, gad, bed, d(rm0), d(rv0), torch.zeros((), dtype=torch.int64, device=dev), training=True, act_slope=0.3)
    assert rel(y2, y[:per]) < 1e-5
    # eval mode: running statistics, no tape
    ye, ste = Lm.wav_front_fwd(d(audio), wd, bd, gad, bed, rm_d, rv_d, nbt, stride=stride, pad=pad, training=False, groups=groups)
    ce = F.conv1d(audio.double().unsqueeze(1), P[0], P[1], stride=stride, padding=pad)
    ye_ref = F.leaky_relu(F.batch_norm(ce, rm_d.double().cpu(), rv_d.double().cpu(), P[2], P[3], training=False, eps=1e-5), 0.3)
    assert rel(ye, cl(ye_ref)) < 1e-5 and ste.gate is None and int(nbt) == groups


# ------------------------------------------------------------------------------------------------ pre-split (bf16 x 3 planes) GEMM
def test_split3_planes_are_exact(pkg, dev):
    """hi + mid + lo == x bit for bit, every term a bf16; padding columns and the extra row are zero; the memory is slab-tiled
    ([cwp / 32][rows + 1][32] per plane: include/trimodal_hip.h)."""
    ops = pkg.ops
    x = (rnd(37, 108, seed=51) * torch.logspace(-6, 6, 108)).to(dev)
    pl = ops.split3_planes(x)
    assert pl.cwp == 128 and pl.t.numel() == 3 * 38 * 128
    ev = pl.element_view()                                                 # [3][rows + 1][cwp] in matrix order
    s = ev[0].double() + ev[1].double() + ev[2].double()
    assert torch.equal(s[:37, :108].float(), x) and float(s[:37, 108:].abs().max()) == 0 and float(s[37].abs().max()) == 0
    # the layout itself: slab 2 (columns 64..95) of row 5 sits at ((2 * 38 + 5) * 32) elements into the plane
    raw = pl.t.view(3, -1)
    assert torch.equal(raw[0, (2 * 38 + 5) * 32:(2 * 38 + 5) * 32 + 32].double() + raw[1, (2 * 38 + 5) * 32:(2 * 38 + 5) * 32 + 32].double()
                       + raw[2, (2 * 38 + 5) * 32:(2 * 38 + 5) * 32 + 32].double(), x[5, 64:96].double())
    xs = torch.zeros(50, 200, device=dev); xs[:, 3:111] = rnd(50, 108, seed=52).to(dev)
    pv = ops.split3_planes(xs[:, 3:111]).element_view()                    # unaligned strided view: scalar path
    assert torch.equal((pv[0].double() + pv[1].double() + pv[2].double())[:50, :108].float(), xs[:, 3:111])


def test_split2h_planes_and_row_scales(pkg, dev):
    """fp16 x 2 planes (csrc/planes.hip tg_split2h_planes): every row scaled by the power of two that puts its largest magnitude into
    [2^14, 2^15); (hi + lo) * inv reproduces x to 2^-22 of the row's largest (fp16 hi and lo, round to nearest); padding columns and the
    extra row are zero; same slab tiling as the bf16 x 3 planes.  Rows spanning 30 decades, a zero row, an unaligned strided source.
    tg_h2_row_scales / tg_win_row_absmax: the product rows' scales of plain and tap windows (zero padding does not count), computed from the
    tensor and from its per-source-row magnitudes alike."""
    ops, Win = pkg.ops, pkg.ops.Win
    x = rnd(37, 108, seed=51) * torch.logspace(-15, 15, 37)[:, None]
    x[11] = 0
    xd = x.to(dev)
    pl = ops.split2h_planes(xd)
    assert pl.kind == "h2" and pl.cwp == 128 and pl.t.shape == (2, 38, 128) and pl.t.dtype == torch.float16 and pl.inv.numel() >= 38
    ev = pl.element_view().double().cpu()
    inv = pl.inv[:38].double().cpu()
    rowmax = x.double().abs().amax(dim=1)
    back = (ev[0] + ev[1]) * inv[:, None]
    assert float(((back[:37, :108] - x.double()).abs() / rowmax.clamp_min(1e-300)[:, None]).max()) <= 2.0 ** -22
    assert float(back[:37, 108:].abs().max()) == 0 and float(back[37].abs().max()) == 0 and float(inv[37]) == 0 and float(back[11].abs().max()) == 0
    live = rowmax > 0
    scaled_max = (ev[0] + ev[1]).abs().amax(dim=1)[:37][live]
    assert float(scaled_max.min()) >= 2.0 ** 14 and float(scaled_max.max()) <= 2.0 ** 15           # the row's largest sits in [2^14, 2^15]
    assert bool(((torch.log2(inv[:37][live]) % 1) == 0).all())                                     # exact powers of two
    raw = pl.t.view(2, -1).double().cpu()
    o = (2 * 38 + 5) * 32                                                                          # slab 2 (columns 64..95) of row 5
    assert float(((raw[0, o:o + 32] + raw[1, o:o + 32]) * inv[5] - x[5, 64:96].double()).abs().max()) <= 2.0 ** -22 * float(rowmax[5])
    xs = torch.zeros(50, 200, device=dev); xs[:, 3:111] = rnd(50, 108, seed=52).to(dev)
    pu = ops.split2h_planes(xs[:, 3:111])                                                          # unaligned strided view: scalar path
    bu = (pu.element_view()[0].double() + pu.element_view()[1].double()) * pu.inv[:51].double()[:, None]
    assert float((bu[:50, :108] - xs[:, 3:111].double()).abs().max()) <= 2.0 ** -22 * float(xs.abs().max())
    # product-row scales: plain rows, and a dilated causal conv window whose early rows reach into the zero padding
    sc = ops.h2_row_scales(Win.plain(xd)).double().cpu()
    assert torch.equal(sc[live] * inv[:37][live], torch.ones(int(live.sum()), dtype=torch.float64))
    B, T, Cc, d = 5, 34, 20, 4
    a = (rnd(B, T, Cc, seed=53) * torch.logspace(-3, 3, T)[None, :, None]).to(dev)
    win = Win.conv(a, 2, pad=d, dil=d, rows_out=T)
    s1 = ops.h2_row_scales(win)
    s2 = ops.h2_row_scales(win, src_rowmax=ops.win_row_absmax(win))
    am = a.double().abs().amax(dim=2).cpu()                                                        # (B, T)
    pad = torch.cat([torch.zeros(B, d, dtype=torch.float64), am], dim=1)
    want_max = torch.maximum(pad[:, :T], pad[:, d:d + T]).reshape(-1)                              # taps t - d and t
    assert torch.equal(s1, s2)
    prod = want_max * s1.double().cpu()
    assert float(prod.min()) >= 2.0 ** 14 and float(prod.max()) < 2.0 ** 15


@pytest.mark.parametrize("B,T1", [(3, 217), (5, 1313), (2, 7891)])
def test_wav_conv2_weight_gradient(pkg, dev, B, T1):
    """Specialised weight / bias gradient of Conv1d(16, 32, 15, stride 6) (csrc/audio.hip) vs torch fp64 autograd; accumulates; frame counts
    that leave a partial group of four output frames."""
    ops = pkg.ops
    x = rnd(B, 16, T1, seed=71)
    w = rnd(32, 16, 15, seed=72, scale=0.1).double().requires_grad_(True)
    b = rnd(32, seed=73).double().requires_grad_(True)
    y = F.conv1d(x.double(), w, b, stride=6)
    dy = rnd(B, 32, y.shape[2], seed=74)
    gw, gb = torch.autograd.grad(y, (w, b), dy.double())
    dW, db = torch.zeros(32, 16, 15, device=dev), torch.zeros(32, device=dev)
    ops.wav_conv2_wgrad(cl(dy).to(dev), cl(x).to(dev), dW, db)
    assert rel(dW, gw) < 1e-4 and rel(db, gb) < 1e-4, (rel(dW, gw), rel(db, gb))
    ops.wav_conv2_wgrad(cl(dy).to(dev), cl(x).to(dev), dW, None)
    assert rel(dW, 2 * gw) < 1e-4 and rel(db, gb) < 1e-4


# ------------------------------------------------------------------------------------------------ fused small sub-networks
def test_speaker_path_fused_forward_backward(pkg, dev):
    """tg_speaker_fwd / tg_speaker_bwd (model/multimodal_context_net.py:83-95,125-137; embedding_net.py:10-13) vs torch fp64 autograd:
    embedding -> Linear(16,16) -> mu / logvar -> z = mu + eps exp(0.5 logvar) -> repeated over the frames; duplicate speaker ids; direct
    gradients on mu / logvar (the KLD term)."""
    ops = pkg.ops
    S, B, T, W = 11, 21, 5, 24
    table, w1, b1 = rnd(S, 16, seed=91), rnd(16, 16, seed=92, scale=0.3), rnd(16, seed=93, scale=0.1)
    wmu, bmu, wlv, blv = rnd(16, 16, seed=94, scale=0.3), rnd(16, seed=95, scale=0.1), rnd(16, 16, seed=96, scale=0.3), rnd(16, seed=97, scale=0.1)
    eps = rnd(B, 16, seed=98)
    vid = torch.randint(0, S, (B,), generator=torch.Generator().manual_seed(99))
    P = [t.double().requires_grad_(True) for t in (table, w1, b1, wmu, bmu, wlv, blv)]
    se_r = P[0][vid]
    zc_r = se_r @ P[1].t() + P[2]
    mu_r, lv_r = zc_r @ P[3].t() + P[4], zc_r @ P[5].t() + P[6]
    z_r = mu_r + eps.double() * torch.exp(0.5 * lv_r)
    d = lambda t: t.to(dev).contiguous()
    rep = torch.full((B * T, W), float("nan"), device=dev)
    se, zc, mu, lv, z = ops.speaker_fwd(d(table), d(vid), d(w1), d(b1), d(wmu), d(bmu), d(wlv), d(blv), d(eps), rep=rep[:, 5:21], T=T)
    for got, ref in ((se, se_r), (zc, zc_r), (mu, mu_r), (lv, lv_r), (z, z_r)):
        assert rel(got, ref) < 1e-5
    assert torch.equal(rep[:, 5:21].view(B, T, 16), z[:, None, :].expand(B, T, 16)) and bool(torch.isnan(rep[:, :5]).all()) and bool(torch.isnan(rep[:, 21:]).all())
    # eps drawn inside the launch (torch.randn_like of :92) == tg_normal with the same state and site, then the explicit-eps form: bit for bit
    st = ops.new_rng_state(77, dev)
    eps_ref = ops.normal(torch.empty(B, 16, device=dev), st, 4)
    eps_out = torch.full((B, 16), float("nan"), device=dev)
    drawn = ops.speaker_fwd(d(table), d(vid), d(w1), d(b1), d(wmu), d(bmu), d(wlv), d(blv), eps_out, draw=(st, 4))
    given = ops.speaker_fwd(d(table), d(vid), d(w1), d(b1), d(wmu), d(bmu), d(wlv), d(blv), eps_ref)
    assert torch.equal(eps_out, eps_ref) and all(torch.equal(a, b_) for a, b_ in zip(drawn, given))
    dz, dmu_in, dlv_in = rnd(B, 16, seed=100), rnd(B, 16, seed=101, scale=0.1), rnd(B, 16, seed=102, scale=0.1)
    loss = (z_r * dz.double()).sum() + (mu_r * dmu_in.double()).sum() + (lv_r * dlv_in.double()).sum()
    grads = torch.autograd.grad(loss, P, retain_graph=True)
    G = [torch.zeros_like(t, device=dev) for t in (table, w1, b1, wmu, bmu, wlv, blv)]
    ops.speaker_bwd(d(dz), d(dmu_in), d(dlv_in), lv, d(eps), zc, se, d(vid), d(w1), d(wmu), d(wlv), G[1], G[2], G[3], G[4], G[5], G[6], G[0])
    for got, ref in zip(G, grads):
        assert rel(got, ref) < 1e-4, (got.shape, rel(got, ref))
    ops.speaker_bwd(d(dz), None, None, lv, d(eps), zc, se, d(vid), d(w1), d(wmu), d(wlv), G[1], G[2], G[3], G[4], G[5], G[6], G[0])   # accumulates; no direct grads
    g2 = torch.autograd.grad((z_r * dz.double()).sum(), P)
    for got, r1, r2 in zip(G, grads, g2):
        assert rel(got, r1 + r2) < 1e-4


@pytest.mark.parametrize("H,Hm,D,M", [(300, 150, 27, 400), (8, 4, 27, 50)])
def test_output_mlp_as_one_linear_map(pkg, dev, H, Hm, D, M):
    """tg_out_mlp_compose / tg_out_mlp_param_grads: Linear(H, Hm) -> LeakyReLU(True) (identity) -> Linear(Hm, D) through the composed weight
    (model/multimodal_context_net.py:100-104) vs torch fp64 autograd of the two-layer form."""
    ops, Win = pkg.ops, pkg.ops.Win
    w1, b1, w2, b2 = rnd(Hm, H, seed=111, scale=0.1), rnd(Hm, seed=112, scale=0.1), rnd(D, Hm, seed=113, scale=0.1), rnd(D, seed=114, scale=0.1)
    o, d_out = rnd(M, H, seed=115), rnd(M, D, seed=116)
    P = [t.double().requires_grad_(True) for t in (w1, b1, w2, b2)]
    od = o.double().requires_grad_(True)
    out_r = F.leaky_relu(od @ P[0].t() + P[1], 1.0) @ P[2].t() + P[3]
    grads = torch.autograd.grad(out_r, P + [od], d_out.double())
    d = lambda t: t.to(dev).contiguous()
    w21, w21t, b21 = ops.out_mlp_compose(d(w1), d(b1), d(w2), d(b2))
    assert torch.equal(w21.t().contiguous(), w21t)
    out = ops.gemm_nt(Win.plain(d(o)), w21, b21, torch.empty(M, D, device=dev))
    assert rel(out, out_r) < 1e-5
    Pm, sv = torch.zeros(D, H, device=dev), torch.zeros(D, device=dev)
    ops.gemm_tn(d(d_out), Win.plain(d(o)), Pm, dbias=sv)
    G = [torch.zeros_like(t, device=dev) for t in (w1, b1, w2, b2)]
    ops.out_mlp_param_grads(Pm, sv, d(w1), d(b1), d(w2), G[0], G[1], G[2], G[3])
    for got, ref in zip(G, grads[:4]):
        assert rel(got, ref) < 1e-4, (got.shape, rel(got, ref))
    do = ops.gemm_nt(Win.plain(d(d_out)), w21t, None, torch.empty(M, H, device=dev))
    assert rel(do, grads[4]) < 1e-4
    # dup = 2: the same map on a bidirectional input [fwd | rev] (o = fwd + rev is never formed)
    yf, yr = rnd(M, H, seed=117), o - rnd(M, H, seed=117)
    y2 = torch.cat([yf, yr], 1)
    w21d, w21td, b21d = ops.out_mlp_compose(d(w1), d(b1), d(w2), d(b2), dup=2)
    assert rel(ops.gemm_nt(Win.plain(d(y2)), w21d, b21d, torch.empty(M, D, device=dev)), out_r) < 1e-5
    P2, s2 = torch.zeros(D, 2 * H, device=dev), torch.zeros(D, device=dev)
    ops.gemm_tn(d(d_out), Win.plain(d(y2)), P2, dbias=s2)
    G2 = [torch.zeros_like(t, device=dev) for t in (w1, b1, w2, b2)]
    ops.out_mlp_param_grads(P2, s2, d(w1), d(b1), d(w2), G2[0], G2[1], G2[2], G2[3], dup=2)
    for got, ref in zip(G2, grads[:4]):
        assert rel(got, ref) < 1e-4
    dy2 = ops.gemm_nt(Win.plain(d(d_out)), w21td, None, torch.empty(M, 2 * H, device=dev))
    assert rel(dy2[:, :H], grads[4]) < 1e-4 and torch.equal(dy2[:, :H], dy2[:, H:])


def test_iter_begin_advances_counters(pkg, dev):
    ops = pkg.ops
    ra, rb = ops.new_rng_state(5, dev), ops.new_rng_state(6, dev)
    ca, cb = torch.full((), 3, dtype=torch.int32, device=dev), torch.full((), 9, dtype=torch.int32, device=dev)
    ops.iter_begin(ra, rb, ca, None)
    ops.iter_begin(ra, None, ca, cb)
    assert ra.tolist() == [5, 2] and rb.tolist() == [6, 1] and int(ca) == 5 and int(cb) == 10


# ------------------------------------------------------------------------------------------------ fp16 x 2 operand interface (ABI 7)
def test_fp16x2_row_magnitudes_through_a_conv_chain(pkg, dev):
    """The magnitude side channel of the fp16 x 2 products (tg_gemm_nt_problem.a_rowmax / c_rowmax / c2_rowmax): a dilated causal conv with
    ReLU, dropout scale and the residual second output (model/tcn.py:27-46) reads its input's row magnitudes (tg_win_row_absmax) and leaves
    EXACTLY the row magnitudes of both tensors it wrote -- the next conv of the chain is then scaled without a pass over its operand and
    agrees with fp64 like the first."""
    ops, Win, Lm = pkg.ops, pkg.ops.Win, pkg.layers
    B, T, Cc, d = 384, 34, 300, 2
    g = torch.Generator().manual_seed(61)
    x = (torch.randn(B, T, Cc, generator=g) * torch.pow(10.0, torch.randint(-3, 3, (B, T, 1), generator=g).float())).to(dev)
    w1, w2 = ((torch.randn(Cc, 2 * Cc, generator=g) * 0.05).to(dev) for _ in range(2))
    b1 = torch.randn(Cc, generator=g).to(dev)
    mask = ((torch.rand(B, T, Cc, generator=g) > 0.3).float() / 0.7).to(dev)
    res = torch.randn(B, T, Cc, generator=g).to(dev)
    pl1, pl2 = ops.split2h_planes(w1), ops.split2h_planes(w2)
    rm = ops.zeros(3, B * T, device=dev)
    a1 = Win.conv(x, 2, pad=d, dil=d, rows_out=T)
    ops.win_row_absmax(a1, out=rm[0])
    assert torch.equal(rm[0].view(B, T), x.abs().amax(dim=2))
    o, o2 = torch.empty(B, T, Cc, device=dev), torch.empty(B, T, Cc, device=dev)
    p1 = [dict(A=a1, W=w1, bias=b1, out=o, act_slope=0.0, out_scale=mask, res=res, out2=o2, res_slope=0.0, c_batch_stride=o.stride(0),
               c_row_stride=o.stride(1), c_rows_out=T, w_planes=pl1, a_rowmax=rm[0], out_rowmax=rm[1], out2_rowmax=rm[2])]
    assert ops.nt_kernel_plan(p1)[0] == 2
    ops.gemm_nt_group(p1)
    xp = torch.cat([torch.zeros(B, d, Cc, dtype=torch.float64, device=dev), x.double()], dim=1)
    ref = torch.relu(xp[:, :T] @ w1[:, :Cc].double().t() + xp[:, d:d + T] @ w1[:, Cc:].double().t() + b1.double()) * mask.double()
    rowsc = ref.abs().amax(dim=2, keepdim=True).clamp_min(1e-30)
    assert float(((o.double() - ref).abs() / rowsc).max()) < 1e-5
    assert torch.equal(rm[1].view(B, T), o.abs().amax(dim=2)) and torch.equal(rm[2].view(B, T), o2.abs().amax(dim=2))      # exact: maxima of what was stored
    # the next conv of the chain: scaled by what the first one left
    o3 = torch.empty(B, T, Cc, device=dev)
    a2 = Win.conv(o2, 2, pad=2 * d, dil=2 * d, rows_out=T)
    p2 = [dict(A=a2, W=w2, bias=None, out=o3, c_batch_stride=o3.stride(0), c_row_stride=o3.stride(1), c_rows_out=T, w_planes=pl2, a_rowmax=rm[2])]
    assert ops.nt_kernel_plan(p2)[0] == 2
    ops.gemm_nt_group(p2)
    o2p = torch.cat([torch.zeros(B, 2 * d, Cc, dtype=torch.float64, device=dev), o2.double()], dim=1)
    ref3 = o2p[:, :T] @ w2[:, :Cc].double().t() + o2p[:, 2 * d:2 * d + T] @ w2[:, Cc:].double().t()
    assert float(((o3.double() - ref3).abs() / ref3.abs().amax(dim=2, keepdim=True).clamp_min(1e-30)).max()) < 1e-5
    # refused where no kernel produces them
    small = torch.empty(64, 48, device=dev)
    with pytest.raises(Exception):
        ops.gemm_nt(Win.plain(torch.randn(64, 64, device=dev)), torch.randn(48, 64, device=dev), None, small, out_rowmax=torch.zeros(64, device=dev))


def test_fp16x2_gru_input_gradient_form(pkg, dev):
    """The GRU layer's input gradient as layers.gru_stack_bwd issues it (multimodal_context_net.py:98-99 backward): dx = [dgi_fwd | dgi_rev] @
    [W_ih_fwd ; W_ih_rev] as ONE product over K = 6H -- two taps of one window, the weight operand the K-concatenated TRANSPOSE of the two
    parameters as fp16 x 2 planes (tg_split2h_planes_tcat: bit-identical to splitting the explicit concatenation), one power-of-two scale per
    CLIP of T rows (a_rowmax_rows = T: what tg_gru_backward_cluster_stats leaves), on the mover-wave kernel's 128 x 96 tile; clips spanning six
    decades; against fp64 at the fp32 tolerance, and the dropout scale of the layer below in the epilogue."""
    ops, Win = pkg.ops, pkg.ops.Win
    nb, T, H, Kin = 128, 34, 300, 600
    M = nb * T
    g = torch.Generator().manual_seed(62)
    dgi = (torch.randn(2, nb, T, 3 * H, generator=g) * torch.pow(10.0, torch.randint(-7, -1, (2, nb, 1, 1), generator=g).float())).to(dev)
    w_f, w_r = ((torch.randn(3 * H, Kin, generator=g) * 0.05).to(dev) for _ in range(2))
    pl = ops.split2h_planes_tcat(w_f, w_r)
    cat = torch.cat([w_f.t(), w_r.t()], dim=1).contiguous()                                        # [Kin][6H]
    pl_ref = ops.split2h_planes(cat)
    assert torch.equal(pl.t, pl_ref.t) and torch.equal(pl.inv[:Kin + 1], pl_ref.inv[:Kin + 1])
    clipmax = dgi.abs().amax(dim=(2, 3)).contiguous()                                              # [2][nb]
    wt = torch.stack([w_f.t().contiguous(), w_r.t().contiguous()])                                 # the fp32 segments the other kernels read
    a_cat = Win(dgi, batches=1, batch_stride=0, row_stride=3 * H, rows_in=2 * M, rows_out=M, cw=3 * H, K=6 * H, dil=M)
    mask = ((torch.rand(M, Kin, generator=g) > 0.3).float() / 0.7).to(dev)
    dx = torch.empty(M, Kin, device=dev)
    kw = dict(b_seg=(3 * H, Kin * 3 * H), out_scale=mask, w_planes=pl, a_rowmax=clipmax.view(-1), a_rowmax_rows=T)
    assert ops.nt_kernel_plan([dict(A=a_cat, W=wt[0], bias=None, out=dx, **kw)]) == (2, 128, 96)
    ops.gemm_nt(a_cat, wt[0], None, dx, **kw)
    ref = (dgi[0].reshape(M, -1).double() @ w_f.double() + dgi[1].reshape(M, -1).double() @ w_r.double()) * mask.double()
    err = (dx.double() - ref).abs() / ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
    assert float(err.max()) < 1e-5, float(err.max())


def test_gru_backward_cluster_leaves_magnitudes(pkg, dev):
    """tg_gru_backward_cluster_stats: the backward recurrence leaves max |dgi| per batch row (over all T steps) and the column maxima of dgi and
    dgh -- exactly the maxima of what it stored (running maxima in registers, one atomic per row / column when the kernel exits); the
    gradients themselves are those of the plain entry point bit for bit.  Row chunks (B = 256 backward = 2 x 128) accumulate."""
    ops = pkg.ops
    T, H = 34, 300
    for B in (128, 256):
        g = torch.Generator().manual_seed(B + 3)
        w = [(torch.randn(3 * H, H, generator=g) * 0.08).to(dev) for _ in range(2)]
        b = [(torch.randn(3 * H, generator=g) * 0.05).to(dev) for _ in range(2)]
        wt = [x.t().contiguous() for x in w]
        gi = (torch.randn(2, B, T, 3 * H, generator=g) * 0.5).to(dev)
        dy = (torch.randn(B, T, 2 * H, generator=g) * torch.pow(10.0, torch.randint(-6, 0, (B, 1, 1), generator=g).float())).to(dev)
        y, sv = torch.empty(B, T, 2 * H, device=dev), torch.empty(2, B, T, 4 * H, device=dev)
        ops.gru_forward(gi, w, b, y, sv)
        out = []
        for with_stats in (False, True):
            dgi, dgh = torch.empty(2, B, T, 3 * H, device=dev), torch.empty(2, B, T, 3 * H, device=dev)
            stats = (ops.zeros(2, B, device=dev), ops.zeros(2, 3 * H, device=dev), ops.zeros(2, 3 * H, device=dev)) if with_stats else None
            filled = ops.gru_backward(dy, y, sv, wt, dgi, dgh, torch.zeros(4 * B * H, device=dev), stats=stats)
            assert bool(filled) == with_stats
            out.append((dgi, dgh, stats))
        ops.check_async_errors()
        (dgi0, dgh0, _), (dgi1, dgh1, (rm, ci, ch)) = out
        assert torch.equal(dgi0, dgi1) and torch.equal(dgh0, dgh1)
        assert torch.equal(rm, dgi1.abs().amax(dim=(2, 3)))
        assert torch.equal(ci, dgi1.abs().amax(dim=(1, 2))) and torch.equal(ch, dgh1.abs().amax(dim=(1, 2)))


def test_absmax_rows_cols(pkg, dev):
    """tg_absmax_rows_cols: row and per-group column magnitudes of a strided matrix in one pass, exact."""
    ops = pkg.ops
    g = torch.Generator().manual_seed(64)
    big = (torch.randn(2 * 1234, 1000, generator=g) * torch.logspace(-5, 5, 1000)).to(dev)
    x = big[:, 52:52 + 900]                                                                      # row stride 1000, 16-byte aligned start
    rm, cm = ops.absmax_rows_cols(x, groups=2, want_rows=True)
    assert torch.equal(rm, x.abs().amax(dim=1)) and torch.equal(cm, x.view(2, 1234, 900).abs().amax(dim=1))
    _, cm1 = ops.absmax_rows_cols(x)
    assert torch.equal(cm1.view(-1), x.abs().amax(dim=0))


# ------------------------------------------------------------------------------------------------ few-row inference recurrence (ABI 8)
@pytest.mark.parametrize("H", [300, 128, 320])
def test_gru_vec_inference_recurrence(pkg, dev, H):
    """csrc/gru_vec.hip (1 <= B <= 4 sequences, no saved gates) against torch.nn.GRU in fp64 and against the cluster kernels, launch after
    launch on ONE workspace: the two exchange buffers alternate between launches, the batch size (row count template) and T change in
    between, so every launch must find its buffer all-sentinel whatever the previous one left behind.  A stale or missed hand-off word would
    show as an O(1e-2) error; the kernel's arithmetic is plain fp32."""
    ops = pkg.ops
    g = torch.Generator().manual_seed(H)
    gru = torch.nn.GRU(8, H, num_layers=1, batch_first=True, bidirectional=True).double()
    w = [gru.weight_hh_l0.detach().float().to(dev).contiguous(), gru.weight_hh_l0_reverse.detach().float().to(dev).contiguous()]
    b = [gru.bias_hh_l0.detach().float().to(dev).contiguous(), gru.bias_hh_l0_reverse.detach().float().to(dev).contiguous()]
    wi = [gru.weight_ih_l0.detach(), gru.weight_ih_l0_reverse.detach()]
    bi = [gru.bias_ih_l0.detach(), gru.bias_ih_l0_reverse.detach()]
    assert ops.GRU_VEC
    for B, T in [(1, 34), (4, 34), (2, 7), (1, 2), (3, 34), (1, 1), (1, 34), (4, 5), (1, 34), (1, 34)]:
        assert ops.gru_vec_takes(B, H, None, None)
        x = torch.randn(B, T, 8, generator=g, dtype=torch.float64)
        with torch.no_grad():
            ref, _ = gru(x)
        gi = torch.stack([x @ wi[d].t() + bi[d] for d in range(2)]).float().to(dev).contiguous()
        y = torch.full((B, T, 2 * H), float("nan"), device=dev)
        ops.gru_forward(gi, w, b, y, None)
        ops.check_async_errors()
        assert bool(torch.isfinite(y).all()), (B, T)
        assert float((y.double().cpu() - ref).abs().max()) < 5e-6, (B, T, float((y.double().cpu() - ref).abs().max()))
        if H == 300 and T == 34:
            prev = ops.GRU_VEC
            ops.GRU_VEC = False
            try:
                yc = ops.gru_forward(gi, w, b, torch.empty_like(y), None)
            finally:
                ops.GRU_VEC = prev
            assert float((y - yc).abs().max()) < 3e-6
    # not taken: more rows, saved gates
    assert not ops.gru_vec_takes(5, 300, None, None) and not ops.gru_vec_takes(1, 300, torch.empty(1), None) and not ops.gru_vec_takes(1, 64, None, None)


def test_gru_backward_cluster_fp16x2_on_rows_of_very_different_scale(pkg, dev):
    """The backward cluster recurrence exchanges a GRADIENT tile as fp16 x 2 planes: its producer scales every (batch row, member) block by the
    block's own power of two and ships the exponent inside the lo plane (csrc/gru_cluster_x3.hip, xc_take_exp).  Rows of one launch whose
    gradients differ by six decades -- what a mean over clips of different loss scale hands back -- must each come out at fp32 accuracy
    RELATIVE TO THAT ROW (a shared scale would lose the small rows entirely, a wrong exponent bit shows as a factor of two), against an fp64
    restatement of the recurrence on the same taped forward (multimodal_context_net.py:155, nn.GRU backward)."""
    ops = pkg.ops
    T, H = 34, 300
    for B, scale in ((128, 1.0), (37, 1e-3), (64, 1e3)):
        g = torch.Generator().manual_seed(11 + B)
        gi = (torch.randn(2, B, T, 3 * H, generator=g) * 0.5).to(dev)
        w = [(torch.randn(3 * H, H, generator=g) * 0.08).to(dev) for _ in range(2)]
        b = [(torch.randn(3 * H, generator=g) * 0.05).to(dev) for _ in range(2)]
        y = torch.empty(B, T, 2 * H, device=dev); sv = torch.empty(2, B, T, 4 * H, device=dev)
        ops.gru_forward(gi, w, b, y, sv)
        dy = (torch.randn(B, T, 2 * H, generator=g) * scale * torch.logspace(-5, 1, B).view(B, 1, 1)).to(dev)
        dy[B // 2] = 0.0                                             # a row without any gradient: all-zero blocks
        wt = [x.t().contiguous() for x in w]
        dgi = torch.full((2, B, T, 3 * H), float("nan"), device=dev); dgh = torch.full_like(dgi, float("nan"))
        ops.gru_backward(dy, y, sv, wt, dgi, dgh, torch.zeros(4 * B * H, device=dev))
        ops.check_async_errors()
        assert ops.gru_cluster_chunks(B, H, bwd=True) is not None     # the cluster kernel is what ran
        # fp64 restatement from the taped gates (sv = r, z, n, W_hn h + b_hn per step)
        r_gi, r_gh = [], []
        for d in range(2):
            W = w[d].double()
            yd, s, dyd = y[..., d * H:(d + 1) * H].double(), sv[d].double(), dy[..., d * H:(d + 1) * H].double()
            dh = torch.zeros(B, H, dtype=torch.float64, device=dev)
            a, c = torch.zeros(B, T, 3 * H, dtype=torch.float64, device=dev), torch.zeros(B, T, 3 * H, dtype=torch.float64, device=dev)
            for t in (range(T - 1, -1, -1) if d == 0 else range(T)):
                tp = t - 1 if d == 0 else t + 1
                hp = yd[:, tp] if 0 <= tp < T else torch.zeros(B, H, dtype=torch.float64, device=dev)
                r, z, n, hn = s[:, t, :H], s[:, t, H:2 * H], s[:, t, 2 * H:3 * H], s[:, t, 3 * H:]
                dht = dyd[:, t] + dh
                dn = dht * (1 - z) * (1 - n * n)
                dz = dht * (hp - n) * z * (1 - z)
                dr = dn * hn * r * (1 - r)
                a[:, t] = torch.cat([dr, dz, dn], 1); c[:, t] = torch.cat([dr, dz, dn * r], 1)
                dh = dht * z + c[:, t] @ W
            r_gi.append(a); r_gh.append(c)
        r_gi, r_gh = torch.stack(r_gi), torch.stack(r_gh)
        assert bool(torch.isfinite(dgi).all()) and bool(torch.isfinite(dgh).all())
        rowmax = r_gi.abs().amax(dim=(0, 2, 3)).view(1, B, 1, 1)
        live = (rowmax > 0).expand_as(r_gi)
        e_gi = float((((dgi.double() - r_gi).abs() / rowmax.clamp_min(1e-300))[live]).max())
        e_gh = float((((dgh.double() - r_gh).abs() / rowmax.clamp_min(1e-300))[live]).max())
        assert e_gi < 1e-6 and e_gh < 1e-6, (B, scale, e_gi, e_gh)
        assert float(dgi[:, B // 2].abs().max()) == 0.0 and float(dgh[:, B // 2].abs().max()) == 0.0
