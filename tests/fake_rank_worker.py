"""Stand-in for a bench rank (tests/test_bench_launcher_cpu.py): checks the environment bench.launch_ranks builds, optionally fails or
hangs on one rank, prints a JSON line on rank 0.  Never touches a GPU."""
import json
import os
import sys
import time

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
mode = os.environ.get("FAKE_MODE", "ok")
if mode == "fail" and rank == world - 1:
    sys.exit(3)
if mode == "fail":
    time.sleep(60)                      # peers of a failed rank wait in a "collective": the launcher must stop them
if mode == "rendezvous":
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)
    assert float(t) == world * (world + 1) / 2
    dist.destroy_process_group()
if rank == 0:
    print("noise before the result line")
    print(json.dumps({"n_gpus": world, "argv": sys.argv[1:]}))
else:
    print(f"rank {rank} chatter that must not reach the parent's stdout")
