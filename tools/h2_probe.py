"""Round 6: fp16 x 2 (three MFMAs per product) against bf16 x 3 (six) on the mover-wave NT kernel -- time per launch (interleaved rounds)
and error against fp64 on 8-decade operands.  Run on the GPU box: python tools/h2_probe.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")


def row_err(out, ref):
    return float(((out.double() - ref).abs() / ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)).max())


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
    print("# tools/h2_probe.py: us per launch (median of 5 x 100), error = max over rows of |out - fp64| / max|fp64 row|")
    for name, M, N, K, nprob, decades in (("gru proj 2 x [13056 x 900 x 600]", 13056, 900, 600, 2, True), ("gru proj 2 x [13056 x 900 x 108]", 13056, 900, 108, 2, True),
                                           ("tcn conv [13056 x 300 x 600]", 13056, 300, 600, 1, True), ("ragged [9999 x 596 x 1000]", 9999, 596, 1000, 1, True),
                                           ("gru proj, unit-range rows", 13056, 900, 600, 2, False)):
        g = torch.Generator().manual_seed(M + N + K)
        x = torch.randn(M, K, generator=g)
        if decades:
            x = x * torch.pow(10.0, torch.randint(-4, 4, (M, 1), generator=g).float())
        xd = x.to(dev)
        ws = [(torch.randn(N, K, generator=g) * 0.1).to(dev) for _ in range(nprob)]
        bs = [torch.randn(N, generator=g).to(dev) for _ in range(nprob)]
        outs = {k: [torch.empty(M, N, device=dev) for _ in range(nprob)] for k in ("x3", "h2")}
        pl3 = [ops.split3_planes(w) for w in ws]
        pl2 = [ops.split2h_planes(w) for w in ws]
        A = Win.plain(xd)
        scale = ops.h2_row_scales(A)
        p3 = [dict(A=A, W=w, bias=b, out=o, act_slope=0.3, w_planes=p) for w, b, o, p in zip(ws, bs, outs["x3"], pl3)]
        p2 = [dict(A=A, W=w, bias=b, out=o, act_slope=0.3, w_planes=p, a_row_scale=scale) for w, b, o, p in zip(ws, bs, outs["h2"], pl2)]
        p2pre = [dict(A=A, W=w, bias=b, out=o, act_slope=0.3, w_planes=p) for w, b, o, p in zip(ws, bs, outs["h2"], pl2)]
        assert ops.nt_kernel_plan(p3)[0] == 2 and ops.nt_kernel_plan(p2)[0] == 2, (ops.nt_kernel_plan(p3), ops.nt_kernel_plan(p2))
        ops.gemm_nt_group(p3); ops.gemm_nt_group(p2)
        torch.cuda.synchronize()
        errs = {}
        for k in ("x3", "h2"):
            errs[k] = max(row_err(o, torch.nn.functional.leaky_relu(xd.double() @ w.double().t() + b.double(), 0.3)) for o, w, b in zip(outs[k], ws, bs))
        t3 = timed(lambda: ops.gemm_nt_group(p3))
        t2 = timed(lambda: ops.gemm_nt_group(p2))
        t3b = timed(lambda: ops.gemm_nt_group(p3))
        t2b = timed(lambda: ops.gemm_nt_group(p2))
        tpre = timed(lambda: ops.h2_row_scales(A, out=scale))
        flop = 2.0 * M * N * K * nprob
        print(f"{name:36s} | x3 {t3:7.1f} / {t3b:7.1f} us ({flop / t3 / 1e6:6.1f} TF) err {errs['x3']:.2e} | h2 {t2:7.1f} / {t2b:7.1f} us ({flop / t2 / 1e6:6.1f} TF) err {errs['h2']:.2e} "
              f"| row-scale pre-pass {tpre:5.1f} us | ratio {t2 / t3:.3f}", flush=True)


if __name__ == "__main__":
    main()
