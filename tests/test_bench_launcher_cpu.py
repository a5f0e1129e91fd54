"""CPU test of `python bench.py --gpus N` started WITHOUT a launcher.

The script must start N ranks itself (torch.distributed.run as a child process), relay exactly
one JSON line from rank 0 and exit with the child's status.  Exercised with the script's --stub
mode (gloo, CPU tensors, a token workload): the launcher, rendezvous and one-line plumbing are
the same as on a GPU node, the attention kernel is not involved.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          env=env, timeout=300)


def test_gpus_2_starts_two_ranks_and_prints_one_json_line():
    p = _run("--gpus", "2", "--stub", "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["verify"]["ranks"] == 2                      # all-reduce of 1 over the ranks
    assert rec["verify"]["heads_by_rank"] == [128, 128]     # the 256 heads of configs[4], split in two
    assert "starting 2 ranks" in p.stderr


def test_world_size_that_disagrees_with_gpus_is_refused():
    # a rank that finds a world other than the --gpus it was asked for must not report a number
    p = _run("--gpus", "4", "--stub", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0",
                                                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert p.returncode == 2
    assert p.stdout.strip() == ""
    assert "refusing" in p.stderr
