"""Does the host run AHEAD of the GPU when the captured iteration is replayed back to back?  Host-side duration of every replay call (no
synchronisation in between) against the iteration's GPU time: if a call returns in a fraction of an iteration, iteration i + 1 is enqueued while
iteration i still runs and the start of an iteration never waits for the host; if a call takes a whole iteration, the graph launch itself waits
for the previous launch of the same executable graph and the head of every iteration is paced by the host's enqueue order.
With --two: two captures of the same step replayed alternately (A, B, A, B ...)."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module("gesture-generation-from-trimodal-context_amd")
pkg._lib.load()
dev = torch.device("cuda:0")
args, G, Dn = bench.build(pkg, dev, seed=0)
trainer = pkg.GanTrainer(G, Dn, args)
text, audio, poses, vid = bench.synthetic_batch(128, 1234, dev)
steps = [pkg.GraphedGanStep(trainer, 20, text, audio, poses, vid, warmup_iters=2)]
if "--two" in sys.argv:
    steps.append(pkg.GraphedGanStep(trainer, 20, text, audio, poses, vid, warmup_iters=0))
for i in range(20): steps[i % len(steps)]()
torch.cuda.synchronize()
n = 64
host = []
t0 = time.perf_counter()
for i in range(n):
    a = time.perf_counter()
    steps[i % len(steps)]()
    host.append(time.perf_counter() - a)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
host.sort()
print(f"{len(steps)} executable graph(s): {n} replays issued in {t_issue * 1e3:.2f} ms of host time, all complete after {t_all * 1e3:.2f} ms "
      f"({t_all / n * 1e3:.3f} ms per iteration); host time per replay call: median {host[n // 2] * 1e3:.3f} ms, min {host[0] * 1e3:.3f}, max {host[-1] * 1e3:.3f}")
