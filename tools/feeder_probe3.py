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
def loop(n, do_put=True, do_ready=True):
    tp = tr_ = ts = 0.0
    feeder.put(*pool[0]); torch.cuda.synchronize()
    t_all = time.perf_counter()
    for k in range(n):
        t0 = time.perf_counter()
        if do_ready: feeder.ready()
        t1 = time.perf_counter()
        if do_put: feeder.put(*pool[(k + 1) % 3])
        t2 = time.perf_counter()
        step()
        t3 = time.perf_counter()
        tr_ += t1 - t0; tp += t2 - t1; ts += t3 - t2
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t_all) / n * 1e3
    print(f"put={do_put} ready={do_ready}: {tot:.2f} ms/iter; host ms/iter: ready {tr_/n*1e3:.2f} put {tp/n*1e3:.2f} step-launch {ts/n*1e3:.2f}", flush=True)
loop(30, False, False); loop(30, True, True); loop(30, True, True)
feeder = data.DeviceBatchFeeder(*step.static, overlap=True)
loop(30, True, True)
