"""Round 6: the backward's input-gradient products on the mover-wave kernel with fp16 x 2 operands, per workgroup tile (TG_MW_TILE=46 / 45 / 43 in
the environment forces 128 x 192 / 128 x 160 / 128 x 96), against the staged-slab bf16 x 3 kernel they run on today.  One process per tile:
    for t in 0 46 45 43; do TG_MW_TILE=$t python tools/h2_dx_probe.py; done"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")


def timed(fn, n=100, rounds=5):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    tile = os.environ.get("TG_MW_TILE", "0")
    g = torch.Generator().manual_seed(3)
    M = 4352
    # GRU input gradient: dx = [dgi_f | dgi_r] @ [W_f ; W_r]: two taps of one window, one [600][1800] weight matrix
    dgi = (torch.randn(2, M, 900, generator=g) * 1e-3).to(dev)
    wcat = (torch.randn(600, 1800, generator=g) * 0.05).to(dev)
    a_cat = Win(dgi, batches=1, batch_stride=0, row_stride=900, rows_in=2 * M, rows_out=M, cw=900, K=1800, dil=M)
    dx = torch.empty(M, 600, device=dev)
    ref = dgi[0].double() @ wcat[:, :900].double().t() + dgi[1].double() @ wcat[:, 900:].double().t()
    rmax = ops.absmax_rows_cols(dgi.view(2 * M, 900), want_rows=True, want_cols=False)[0]
    cases = [("gru dx [4352 x 600 x 1800]", a_cat, wcat, dx, ref, rmax)]
    # text-encoder input gradient: two taps (t, t + d) of dy
    dy = (torch.randn(128, 34, 300, generator=g) * 1e-3).to(dev)
    wt = (torch.randn(300, 600, generator=g) * 0.05).to(dev)
    a_t = Win.taps(dy, 2, shift=0, dil=4, rows_out=34)
    dxt = torch.empty(M, 300, device=dev)
    dyp = torch.cat([dy.double(), torch.zeros(128, 4, 300, dtype=torch.float64, device=dev)], dim=1)
    reft = (dyp[:, :34] @ wt[:, :300].double().t() + dyp[:, 4:38] @ wt[:, 300:].double().t()).reshape(M, 300)
    rmt = ops.absmax_rows_cols(dy.view(M, 300), want_rows=True, want_cols=False)[0]
    cases.append(("tcn dx [4352 x 300 x 600]", a_t, wt, dxt, reft, rmt))
    for name, A, W, out, ref, rm in cases:
        base = lambda: ops.gemm_nt(A, W, None, out)
        base(); torch.cuda.synchronize()
        e0 = float((out.double() - ref).abs().max() / ref.abs().max())
        t0 = timed(base)
        pl = ops.split2h_planes(W)
        prob = [dict(A=A, W=W, bias=None, out=out, w_planes=pl, a_rowmax=rm)]
        plan = ops.nt_kernel_plan(prob)
        if plan[0] != 2:
            print(f"{name:28s} tile {tile}: staged-slab x3 {t0:6.1f} us (err {e0:.1e}) | mover-wave h2: not planned {plan}", flush=True)
            continue
        out.zero_()
        ops.gemm_nt_group(prob); torch.cuda.synchronize()
        e1 = float((out.double() - ref).abs().max() / ref.abs().max())
        t1 = timed(lambda: ops.gemm_nt_group(prob))
        print(f"{name:28s} tile {tile} -> plan {plan}: staged-slab x3 {t0:6.1f} us (err {e0:.1e}) | mover-wave h2 {t1:6.1f} us (err {e1:.1e})", flush=True)
    tp = timed(lambda: ops.absmax_rows_cols(dgi.view(2 * M, 900), want_rows=True, want_cols=True, groups=2))
    print(f"absmax rows + cols over dgi [2 x 4352 x 900]: {tp:.1f} us")


if __name__ == "__main__":
    main()
