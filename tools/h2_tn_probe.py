"""Round 6: the mover-wave weight-gradient kernel with fp16 x 2 operands (per-column scales) against bf16 x 3 -- the four gradients of a GRU layer
at B = 128 and a text-encoder group; us per launch (with and without the column-magnitude pre-pass) and error against fp64."""
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
    g = torch.Generator().manual_seed(1)
    M, H = 4352, 300
    dgi = [(torch.randn(M, 3 * H, generator=g) * torch.pow(10.0, torch.randint(-3, 3, (M, 1), generator=g).float()) * 1e-4).to(dev) for _ in range(2)]
    x = torch.randn(M, 2 * H, generator=g).to(dev)
    hp = torch.randn(M, H, generator=g).to(dev)
    res = {}
    for fmt in ("x3", "h2", "h2pre"):
        ops.GEMM_H2 = ops.TN_AUTO_COLMAX = fmt != "x3"
        probs, refs = [], []
        for d in range(2):
            for A, Kc in ((x, 2 * H), (hp, H)):
                dW, db = torch.zeros(3 * H, Kc, device=dev), torch.zeros(3 * H, device=dev)
                refs.append((dgi[d].double().t() @ A.double(), dgi[d].double().sum(0)))
                probs.append(dict(dY=dgi[d], A=Win.plain(A), dW=dW, dbias=db))
        if fmt == "h2pre":      # column magnitudes supplied: the product alone
            cm = {id(t): ops.absmax_rows_cols(t)[1].view(-1) for t in (dgi[0], dgi[1], x, hp)}
            probs = [dict(p, y_colmax=cm[id(p["dY"])], a_colmax=cm[id(p["A"].t)]) for p in probs]
        assert ops.tn_kernel_plan(probs) == 2
        ops.gemm_tn_group(probs)
        torch.cuda.synchronize()
        e_w = max(float((p["dW"].double() - rw).abs().max() / rw.abs().max()) for p, (rw, rb) in zip(probs, refs))
        e_b = max(float((p["dbias"].double() - rb).abs().max() / rb.abs().max()) for p, (rw, rb) in zip(probs, refs))
        t = timed(lambda: ops.gemm_tn_group(probs))
        res[fmt] = t
        print(f"GRU layer's four weight gradients, {fmt:6s}: {t:7.1f} us per group (product + combine{' + pre-pass' if fmt == 'h2' else ''}), err dW {e_w:.2e} dbias {e_b:.2e}", flush=True)
    tp = timed(lambda: ops.absmax_rows_cols(dgi[0]))
    print(f"absmax pass over one [4352 x 900] operand: {tp:.1f} us")


if __name__ == "__main__":
    main()
