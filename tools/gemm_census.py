"""Per-shape census of the GEMM calls of one training iteration: GPU time per call (HIP events, eager mode), FLOPs, TFLOP/s."""
import importlib, sys, collections, torch
sys.path.insert(0, '/root/repo')
import bench
pkg = importlib.import_module(bench.PKG)
ops = pkg.ops
dev = torch.device("cuda:0")
args, G, D = bench.build(pkg, dev, seed=0)
tr = pkg.GanTrainer(G, D, args)
text, audio, poses, vid = bench.synthetic_batch(128, 1234, dev)
for _ in range(2): tr.train_iter(11, text, audio, poses, vid)
torch.cuda.synchronize()
rec = []
def wrap(name, fn):
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        if name == "nt":
            A, Bw, out = a[0], a[1], a[3]
            M = out.numel() // Bw.shape[0] if out is not None else None
            N, K = Bw.shape[0], A.K
            M = A.M if hasattr(A, "M") else M
        else:
            dy, A, dW = a[0], a[1], a[2]
            M, N, K = dy.shape[0], dy.shape[1], A.K
        rec.append((name, M, N, K, e0, e1))
        return r
    return w
ops.gemm_nt = wrap("nt", ops.gemm_nt)
ops.gemm_tn = wrap("tn", ops.gemm_tn)
for _ in range(3): tr.train_iter(11, text, audio, poses, vid)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for name, M, N, K, e0, e1 in rec:
    a = agg[(name, M, N, K)]; a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
tot = 0
rows = []
for (name, M, N, K), (n, us) in agg.items():
    fl = 2.0 * M * N * K
    rows.append((us / 3, name, M, N, K, n / 3, us / n, fl / (us / n) / 1e6))
rows.sort(reverse=True)
print(f"{'kind':4s} {'M':>8s} {'N':>5s} {'K':>5s} {'calls':>6s} {'us/call':>8s} {'TF':>6s} {'us/iter':>8s} {'us/iter@100TF':>12s}")
t_all = t_ideal = 0
for usi, name, M, N, K, n, usc, tf in rows:
    ideal = 2.0 * M * N * K / 100e6 * n
    t_all += usi; t_ideal += ideal
    print(f"{name:4s} {M:8d} {N:5d} {K:5d} {n:6.1f} {usc:8.1f} {tf:6.1f} {usi:8.1f} {ideal:12.1f}")
print("total us/iter", t_all, "at 100 TF", t_ideal)
