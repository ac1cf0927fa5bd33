"""Round 6: the text encoder's forward convs [13056 x 300 x 600] (stacked forward, two causal taps, fp16 x 2) per forced workgroup tile of the
mover-wave kernel (TG_MW_TILE=46 / 45 / 43 = 128 x 192 / 128 x 160 / 128 x 96; 0 = the menu's own choice).  One process per tile:
    for t in 0 46 45 43; do TG_MW_TILE=$t python tools/h2_conv_tile_probe.py; done
Sustained launches of ONE conv and of the chain of eight as the forward issues them (each conv reads the previous one's output: cold operands)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops, Win = pkg.ops, pkg.ops.Win
dev = torch.device("cuda:0")
B, T, C = 384, 34, 300
g = torch.Generator().manual_seed(5)
xs = [(torch.randn(B, T, C, generator=g) * 0.5).to(dev) for _ in range(9)]
ws = [(torch.randn(C, 2 * C, generator=g) * 0.05).to(dev) for _ in range(8)]
bs = [(torch.randn(C, generator=g) * 0.1).to(dev) for _ in range(8)]
pls = [ops.split2h_planes(w) for w in ws]
rm = [torch.zeros(B * T, device=dev) for _ in range(9)]
ops.win_row_absmax(Win.plain(xs[0].view(B * T, C)), rm[0])
def conv(i, d):
    return dict(A=Win.conv(xs[i], 2, pad=d, dil=d, rows_out=T), W=ws[i], bias=bs[i], out=xs[i + 1], act_slope=0.0, w_planes=pls[i],
                a_rowmax=rm[i], out_rowmax=rm[i + 1], c_batch_stride=xs[i + 1].stride(0), c_row_stride=xs[i + 1].stride(1), c_rows_out=T)
def timed(fn, n=50, rounds=5):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]
one = [conv(0, 4)]
plan = ops.nt_kernel_plan(one)
def chain():
    for i in range(8):
        rm[i + 1].zero_()
        ops.gemm_nt_group([conv(i, 1 << (i // 2))])
ops.gemm_nt_group(one); torch.cuda.synchronize()
t1 = timed(lambda: ops.gemm_nt_group(one))
tz = timed(lambda: [rm[i + 1].zero_() for i in range(8)])
t8 = timed(chain, n=20)
print(f"TG_MW_TILE={os.environ.get('TG_MW_TILE', '0')}: plan {plan}; one conv sustained {t1:6.1f} us; chain of eight {t8:6.1f} us (of which the eight zero fills {tz:5.1f}) = {(t8 - tz) / 8:5.1f} us per conv", flush=True)
