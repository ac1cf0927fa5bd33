import importlib, sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
pkg = importlib.import_module(bench.PKG)
dev = torch.device("cuda:0")
args, G, D = bench.build(pkg, dev, seed=0)
tr = pkg.GanTrainer(G, D, args)
text, audio, poses, vid = bench.synthetic_batch(128, 1234, dev)
step = pkg.GraphedGanStep(tr, 11, text, audio, poses, vid, warmup_iters=2)
h = torch.empty(128, 36267).pin_memory(); d = torch.empty(128, 36267, device=dev)
def ev(): return torch.cuda.Event(enable_timing=True)
def h2d_time(label, before=None, n=5):
    ts = []
    for _ in range(n):
        if before: before()
        e0, e1 = ev(), ev()
        e0.record(); d.copy_(h, non_blocking=True); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{label}: H2D 18.6 MB in ms {['%.2f' % t for t in ts]}", flush=True)
h2d_time("idle")
h2d_time("after graph replay (stream-ordered)", before=lambda: step())
h2d_time("after host rewrite of the pinned buffer", before=lambda: h.mul_(1.0001))
h2d_time("after graph + host rewrite", before=lambda: (step(), h.mul_(1.0001)))
x = torch.randn(4096, 4096, device=dev)
h2d_time("after a plain matmul kernel", before=lambda: torch.mm(x, x))
