"""Few-row inference recurrence (csrc/gru_vec.hip) against the cluster kernel at the same call: time per launch at B = 1 .. 4 (T = 34, H = 300;
HIP events, 200 launches back to back = the decode graph's situation) and the difference of their outputs."""
import importlib, sys, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
T, H = 34, 300
g = torch.Generator().manual_seed(3)
w = [(torch.randn(3 * H, H, generator=g) * 0.08).to(dev) for _ in range(2)]
b = [(torch.randn(3 * H, generator=g) * 0.05).to(dev) for _ in range(2)]
for B in (1, 2, 4):
    gi = (torch.randn(2, B, T, 3 * H, generator=g) * 0.5).to(dev)
    out = {}
    for vec in (False, True):
        ops.GRU_VEC = vec
        y = torch.empty(B, T, 2 * H, device=dev)
        for _ in range(5): ops.gru_forward(gi, w, b, y, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): ops.gru_forward(gi, w, b, y, None)
        e1.record(); e1.synchronize()
        ops.check_async_errors()
        out[vec] = (y.clone(), e0.elapsed_time(e1) / 200 * 1e3)
    d = float((out[True][0] - out[False][0]).abs().max())
    print(f"B={B}: cluster kernel {out[False][1]:6.1f} us per launch ({out[False][1] / T:4.2f} us/step)   few-row kernel {out[True][1]:6.1f} us ({out[True][1] / T:4.2f} us/step)   max |y diff| {d:.2e}", flush=True)
