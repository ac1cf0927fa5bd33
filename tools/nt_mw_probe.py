"""Same-process A/B of the big forward products: mover-wave kernel (csrc/gemm_mw.hip) against the staged-slab kernel
(csrc/gemm_split.hip), interleaved rounds, sustained launches on random data (cdna_hip_programming.md 5.4 rules 24 / 25).

    python3 tools/nt_mw_probe.py [rounds] > gpurun_out/<tag>_nt_mw_probe.txt
Shapes: the stacked forward of the B = 128 iteration (13056 rows): both directions' GRU input projections as one group (K = 600 and the
first layer's K = 108), one TCN conv with its causal window + dropout scale epilogue, and backward-sized products (4352 rows)."""
import importlib
import statistics
import sys

import torch

sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win, Lm = pkg.ops, pkg.ops.Win, pkg.layers
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def group(M, N, K, n):
    x = torch.randn(M, K, device=dev)
    ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(n)]
    bs = [torch.randn(N, device=dev) for _ in range(n)]
    outs = [torch.empty(M, N, device=dev) for _ in range(n)]
    probs = [dict(A=Win.plain(x), W=w, bias=b, out=o, w_planes=ops.split3_planes(w)) for w, b, o in zip(ws, bs, outs)]
    return probs, 2.0 * M * N * K * n, (x, ws, bs, outs)


def tcn(B, T, C, d):
    x = torch.randn(B, T, C, device=dev)
    wp = torch.randn(C, 2 * C, device=dev) * 0.05
    b = torch.randn(C, device=dev)
    mask = (torch.rand(B, T, C, device=dev) > 0.3).float() / 0.7
    out = torch.empty(B, T, C, device=dev)
    probs = [dict(A=Win.conv(x, 2, pad=d, dil=d, rows_out=T), W=wp, bias=b, out=out, act_slope=0.0, out_scale=mask, w_planes=ops.split3_planes(wp),
                  c_batch_stride=out.stride(0), c_row_stride=out.stride(1), c_rows_out=T)]
    return probs, 2.0 * B * T * C * 2 * C, (x, wp, b, mask, out)


cases = [("gru proj 2 x [13056 x 900 x 600]", group(13056, 900, 600, 2)),
         ("gru proj 2 x [13056 x 900 x 108]", group(13056, 900, 108, 2)),
         ("tcn conv  [13056 x 300 x 600] d=4", tcn(384, 34, 300, 4)),
         ("dgrad     [4352 x 600 x 1800]", group(4352, 600, 1800, 1)),
         ("dgrad tcn [4352 x 300 x 600]", group(4352, 300, 600, 1))]
print(f"# tools/nt_mw_probe.py: {rounds} interleaved rounds x 100 launches, us per launch (median / min), fp32-equivalent TFLOP/s at the median")
for name, (probs, flops, keep) in cases:
    res = {}
    plans = {}
    for on in (False, True):
        ops.set_nt_mover_waves(on)
        plans[on] = ops.nt_kernel_plan(probs)
        for _ in range(10):
            ops.gemm_nt_group(probs)
    ref = None
    for r in range(rounds):
        for on in (False, True):
            ops.set_nt_mover_waves(on)
            res.setdefault(on, []).append(timed(lambda: ops.gemm_nt_group(probs), 100))
    ops.set_nt_mover_waves(None)
    line = f"{name:36s}"
    for on in (False, True):
        med, mn = statistics.median(res[on]), min(res[on])
        line += f" | plan {plans[on]} {med:7.1f} / {mn:7.1f} us {flops / med / 1e6:6.1f} TF"
    line += f" | ratio {statistics.median(res[True]) / statistics.median(res[False]):.3f}"
    print(line, flush=True)

# the text encoder's convs as the iteration sees them: operands and outputs of every launch on COLD lines (24 rotating buffer sets, 1.9 GB, past
# the 256 MB Infinity Cache); conv1 = ReLU + dropout scale, conv2 = the same + the block's closing relu(out + x) as second output
NSET = 24
Bc, Tc, Cc = 384, 34, 300
wp = torch.randn(Cc, 2 * Cc, device=dev) * 0.05
bc = torch.randn(Cc, device=dev)
wpl = ops.split3_planes(wp)
sets = [dict(x=torch.randn(Bc, Tc, Cc, device=dev), m=(torch.rand(Bc, Tc, Cc, device=dev) > 0.3).float() / 0.7, r=torch.randn(Bc, Tc, Cc, device=dev),
             o=torch.empty(Bc, Tc, Cc, device=dev), o2=torch.empty(Bc, Tc, Cc, device=dev)) for _ in range(NSET)]
def conv_probs(st, second):
    kw = dict(res=st["r"], out2=st["o2"], res_slope=0.0) if second else {}
    return [dict(A=Win.conv(st["x"], 2, pad=4, dil=4, rows_out=Tc), W=wp, bias=bc, out=st["o"], act_slope=0.0, out_scale=st["m"], w_planes=wpl,
                 c_batch_stride=st["o"].stride(0), c_row_stride=st["o"].stride(1), c_rows_out=Tc, **kw)]
for second in (False, True):
    plist = [conv_probs(st, second) for st in sets]
    def sweep():
        for pr in plist:
            ops.gemm_nt_group(pr)
    sweep()
    ts = [timed(sweep, 4) / NSET for _ in range(rounds)]
    print(f"tcn conv{2 if second else 1} cold  [13056 x 300 x 600] d=4 | plan {ops.nt_kernel_plan(plist[0])} {statistics.median(ts):7.1f} / {min(ts):7.1f} us per launch", flush=True)

# which stream's cold lines cost what (conv1 form = ReLU + dropout scale): every combination of {activation, mask, output} rotating over the 24
# sets (cold) or pinned to set 0 (warm)
print("# cold-line attribution, conv1 form: us per launch with the named streams COLD (rotating buffers), the others warm")
for cold in ((), ("x",), ("m",), ("o",), ("x", "m"), ("x", "o"), ("m", "o"), ("x", "m", "o")):
    plist = []
    for st in sets:
        pick = lambda k: st[k] if k in cold else sets[0][k]
        o = pick("o")
        plist.append([dict(A=Win.conv(pick("x"), 2, pad=4, dil=4, rows_out=Tc), W=wp, bias=bc, out=o, act_slope=0.0, out_scale=pick("m"), w_planes=wpl,
                           c_batch_stride=o.stride(0), c_row_stride=o.stride(1), c_rows_out=Tc)])
    def sweep():
        for pr in plist:
            ops.gemm_nt_group(pr)
    sweep()
    ts = [timed(sweep, 4) / NSET for _ in range(rounds)]
    print(f"  cold: {'+'.join(cold) if cold else 'none':8s} {statistics.median(ts):7.1f} us", flush=True)
# no mask at all (bias + ReLU only), everything cold
plist = [[dict(A=Win.conv(st["x"], 2, pad=4, dil=4, rows_out=Tc), W=wp, bias=bc, out=st["o"], act_slope=0.0, w_planes=wpl,
               c_batch_stride=st["o"].stride(0), c_row_stride=st["o"].stride(1), c_rows_out=Tc)] for st in sets]
def sweep():
    for pr in plist:
        ops.gemm_nt_group(pr)
sweep()
ts = [timed(sweep, 4) / NSET for _ in range(rounds)]
print(f"  no mask operand, x + o cold: {statistics.median(ts):7.1f} us", flush=True)
