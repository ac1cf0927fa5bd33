"""Watches the flag / timeout words of the persistent GRU workspaces while a captured training step is replayed with eager copies
between replays (data.DeviceBatchFeeder).  With hipMemsetAsync in the captured region (the library up to commit \"Persistent cluster
GRU backward\") every replay filled the flag block with the source / destination pointers of the preceding 1 KB copy instead of zeros
(ROCm 7.2, MI355X); with the library's own zero kernel (csrc/common.hpp zero_async) the words stay clean.  N_ITERS / SYNC env vars."""
import importlib, sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
pkg = importlib.import_module(bench.PKG)
data = importlib.import_module(bench.PKG + ".data")
ops = pkg.ops
dev = torch.device("cuda:0")
args, G, D = bench.build(pkg, dev, seed=0)
tr = pkg.GanTrainer(G, D, args)
text, audio, poses, vid = bench.synthetic_batch(128, 1234, dev)
step = pkg.GraphedGanStep(tr, 11, text, audio, poses, vid, warmup_iters=2)
pool = [tuple(t.cpu() for t in bench.synthetic_batch(128, 4321 + i, dev)) for i in range(3)]
pool = [(t, p_, au, v) for (t, au, p_, v) in pool]
feeder = data.DeviceBatchFeeder(*step.static)
names = {}
for k, ws in ops._gru_ws.items(): names[f"ws{k[1:]}"] = ws
for i, t in enumerate(step.static): names[f"static{i}"] = t
for s in range(2):
    for i, t in enumerate(feeder.staging[s]): names[f"staging{s}.{i}"] = t
for n, t in names.items(): print(f"{n:22s} ptr {t.data_ptr():#x} bytes {t.numel() * t.element_size()}")
print("G slab", hex(tr.G.slab.flat.data_ptr()), "grad", hex(tr.G.slab.grad.data_ptr()))
feeder.put(*pool[0])
import os
N = int(os.environ.get("N_ITERS", "40")); SYNC = os.environ.get("SYNC", "0") == "1"
for k in range(N):
    feeder.ready()
    feeder.put(*pool[(k + 1) % 3])
    step()
    if SYNC or k == N - 1: torch.cuda.synchronize()
    else: continue
    for key, ws in ops._gru_ws.items():
        w = ws[:8].tolist()
        if w[0] != 0:
            print("iter", k, key[1:], [hex(x & 0xffffffff) for x in w], "nonzero words in first 4096:", int((ws[:4096] != 0).sum()))
