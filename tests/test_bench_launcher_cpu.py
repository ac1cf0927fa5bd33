"""`python bench.py --gpus N` with no WORLD_SIZE must start its own N ranks (the reference's multi-GPU switch is
scripts/train.py:93-96).  The launcher logic is exercised here with a fake worker: no GPU, no HIP call."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORKER = os.path.join(ROOT, "tests", "fake_rank_worker.py")


@pytest.mark.parametrize("n", [2, 4])
def test_launcher_starts_n_ranks_and_relays_rank0_line(n, capfd):
    import bench
    rc, line = bench.launch_ranks(n, ["--gpus", str(n), "--steps", "3"], worker=WORKER, env_extra={"FAKE_MODE": "rendezvous"})
    assert rc == 0
    out = json.loads(line)
    assert out["n_gpus"] == n and out["argv"] == ["--gpus", str(n), "--steps", "3"]
    cap = capfd.readouterr()
    assert "chatter" not in cap.out and "noise before" not in cap.out        # only the caller prints the relayed line


def test_launcher_reports_failure_and_stops_peers():
    import bench
    t0 = time.time()
    rc, line = bench.launch_ranks(4, [], worker=WORKER, env_extra={"FAKE_MODE": "fail"})
    assert rc == 3 and line is None
    assert time.time() - t0 < 30.0, "peers of the failed rank were not stopped"


def test_bench_cli_becomes_launcher_without_world_size(tmp_path):
    """The real CLI path: bench.py --gpus 2 with WORLD_SIZE unset must not assert; with a mismatching WORLD_SIZE it must refuse
    with a message (not start ranks).  The ranks it starts here have no GPU, so they fail loudly -- what is checked is that the parent
    launched them, relayed the failure as a non-zero exit code and made no GPU call itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "bench launcher: rank" in r.stderr and "needs a GPU" in r.stderr, r.stderr[-2000:]
    env["WORLD_SIZE"] = "4"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
