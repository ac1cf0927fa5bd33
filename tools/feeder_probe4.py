import importlib, sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
pkg = importlib.import_module(bench.PKG)
data = importlib.import_module(bench.PKG + ".data")
dev = torch.device("cuda:0")
args, G, D = bench.build(pkg, dev, seed=0)
tr = pkg.GanTrainer(G, D, args)
text, audio, poses, vid = bench.synthetic_batch(128, 1234, dev)
step = pkg.GraphedGanStep(tr, 11, text, audio, poses, vid, warmup_iters=2)
pool = [tuple(t.cpu() for t in bench.synthetic_batch(128, 4321 + i, dev)) for i in range(3)]
pool = [(t, p_, au, v) for (t, au, p_, v) in pool]
feeder = data.DeviceBatchFeeder(*step.static)
def ev(): return torch.cuda.Event(enable_timing=True)
feeder.put(*pool[0]); torch.cuda.synchronize()
marks = []
for k in range(12):
    e = [ev() for _ in range(3)]
    e[0].record(); feeder.ready(); e[1].record()
    feeder.put(*pool[(k + 1) % 3])
    step(); e[2].record()
    marks.append(e)
torch.cuda.synchronize()
for k, e in enumerate(marks):
    gap = marks[k - 1][2].elapsed_time(e[0]) if k else 0.0
    print(f"iter {k}: gap-before {gap:.2f} ms, ready(copies) {e[0].elapsed_time(e[1]):.2f} ms, graph {e[1].elapsed_time(e[2]):.2f} ms")
