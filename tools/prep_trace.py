"""Round 5: which weight-derived operands (layers.WeightPrep) does each part of the GAN iteration ask for, and from which stream?  Prints, per
cached operand of the generator group, its shape / permutation and the part of the iteration that refreshes it (main0: start, main stream;
side0: head of the forward's forked branch; late: on the forked branch behind the audio encoder), after three eager iterations at the bench
configuration."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module(bench.PKG)
pkg._lib.load()
dev = torch.device("cuda:0")
args, G, Dn = bench.build(pkg, dev, seed=0)
tr = pkg.GanTrainer(G, Dn, args)
text, audio, poses, vid = bench.synthetic_batch(128, 1234, dev)
for _ in range(3):
    tr.train_iter(11, text, audio, poses, vid)
torch.cuda.synchronize()
for gid, g in tr.prep.groups.items():
    print(f"group {gid}: {len(g['jobs'])} operands; tables: " + ", ".join(f"{p} {g['n'].get(p, 0)}" for p in tr.prep.PARTS))
    for src3, dst, perm in g["jobs"]:
        part = tr.prep.part_of[(src3.data_ptr(), tuple(src3.shape), perm)]
        print(f"  {tuple(src3.shape)!s:18} perm {perm!s:12} {part:6} {dst.numel() * dst.element_size() / 1024:8.1f} KB")
