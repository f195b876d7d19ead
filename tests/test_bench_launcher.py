"""bench.py --gpus N without a launcher starts N ranks itself (before anything touches a GPU) with the environment
torch.distributed.run would give them; the ranks meet over the library's TCP communicator. Rehearsed here without a GPU:
`--rehearse-launcher` runs the launcher and the control plane only (stand-in result tables instead of alignments)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env=None, timeout=120):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_gpus_2_spawns_two_ranks_with_the_right_environment():
    r = run(["--gpus", "2", "--steps", "6", "--batch", "5", "--rehearse-launcher"])
    assert r.returncode == 0, r.stderr.decode()
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()   # rank 0's line only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["gathers_ok"] and out["id_ok"]
    assert out["max_over_ranks"] == 2.0          # 1 + rank, maximum over the ranks
    ranks = sorted(out["ranks"], key=lambda e: e["rank"])
    assert [e["rank"] for e in ranks] == [0, 1] and [e["local_rank"] for e in ranks] == [0, 1]
    assert all(e["world_size"] == 2 for e in ranks) and ranks[0]["master_port"] == ranks[1]["master_port"] > 0


def test_three_ranks_and_a_failing_rank_fails_the_launch():
    r = run(["--gpus", "3", "--steps", "4", "--batch", "2", "--rehearse-launcher"])
    assert r.returncode == 0, r.stderr.decode()
    assert json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])["n_gpus"] == 3
    # a rank that cannot run makes the whole launch fail loudly, not hang: the product path with a library that does not exist
    # (deterministic with or without a GPU on the box: the loader fails before anything touches a device)
    r = run(["--gpus", "2", "--steps", "2", "--no-extras", "--lib", "/nonexistent/libellc_hip.so"], timeout=300)
    assert r.returncode != 0


def test_eight_ranks_rehearsal():
    """The launcher at the node's size (configs[3] / configs[4] run with --gpus 8): eight ranks, control plane and pipelined gathers."""
    r = run(["--gpus", "8", "--steps", "4", "--batch", "3", "--rehearse-launcher"], timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["gathers_ok"] and out["id_ok"] and out["max_over_ranks"] == 8.0
    assert sorted(e["rank"] for e in out["ranks"]) == list(range(8))


def test_ranks_started_by_a_launcher_are_not_respawned():
    """WORLD_SIZE in the environment (torch.distributed.run): the process is a rank, not a launcher."""
    r = run(["--gpus", "1", "--steps", "3", "--batch", "2", "--rehearse-launcher"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_PORT": "29871"})
    assert r.returncode == 0, r.stderr.decode()
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["gathers_ok"]
