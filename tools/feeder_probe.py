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
def run(mode, n=12):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); bad = 0
    feeder.put(*pool[0])
    for k in range(n):
        feeder.ready()
        if mode == "overlap":
            feeder.put(*pool[(k + 1) % 3])
        losses = step()
        if mode == "serial":
            torch.cuda.synchronize()
            feeder.put(*pool[(k + 1) % 3])
            feeder.stream.synchronize()
        if mode == "check":
            feeder.put(*pool[(k + 1) % 3])
            torch.cuda.synchronize()
            for key, ws in ops._gru_ws.items():
                if int(ws[0]) != 0 and bad < 3:
                    print("  timeout", key, "step", int(ws[1]), "block", int(ws[2]), "ticks>>10", int(ws[3]), "flags", ws[4:14].tolist(), flush=True)
            try: ops.check_async_errors()
            except RuntimeError as e: bad += 1
    torch.cuda.synchronize()
    try: ops.check_async_errors()
    except RuntimeError as e: bad += 1
    print(mode, f"{(time.perf_counter() - t0) / n * 1e3:.2f} ms/iter", "timeouts", bad, flush=True)
for mode in ("serial", "check"):
    run(mode)
# raw pinned host -> device bandwidth on this box
h = torch.empty(64 << 20, dtype=torch.uint8).pin_memory(); d = torch.empty_like(h, device=dev)
for _ in range(2): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"pinned H2D 64 MiB: {dt * 1e3:.2f} ms = {64 / 1024 / dt:.2f} GiB/s")
hp = torch.empty(64 << 20, dtype=torch.uint8)
t0 = time.perf_counter(); h.copy_(hp); print(f"host memcpy 64 MiB into pinned: {(time.perf_counter() - t0) * 1e3:.2f} ms")
