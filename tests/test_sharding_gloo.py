"""N>1 path on CPU: two gloo ranks shard a batch of alignments and gather the result table, and the same
sharded batch through the product's own gather (sharding.Comm over the TCP transport, libellc_comm.so).
The alignments themselves are produced by a stand-in context (the HIP library needs a GPU); what is under
test is the partition, the padding of uneven shards and the order of the gathered poses."""
import os
import subprocess
import sys
import textwrap
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    sys.path.insert(0, os.path.join(%r, "tests"))
    import numpy as np
    import torch.distributed as dist
    from egomotion_with_local_loop_closures_amd import sharding
    import gloo_gather

    class FakeCtx:
        def __init__(self, lo): self.lo = lo
        def align(self, kf, fr, init_pose=None, mode=0):
            g = self.lo + np.arange(len(kf))
            pose = np.stack([g * 0.5 + k for k in range(6)], axis=1).astype(np.float32)
            iters = np.tile(np.array([[4, 7, 9, 12]]), (len(kf), 1)) + (g[:, None] %% 2)
            return pose, iters, (g * 0.25).astype(np.float32)

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    for total in (8, 7, 1, 5):
        lo, hi = sharding.shard_range(total, world, rank)
        slots = np.arange(hi - lo)
        table = gloo_gather.align_sharded_gloo(FakeCtx(lo), total, slots, slots)
        g = np.arange(total)
        assert table.shape == (total, 8), table.shape
        assert np.array_equal(table[:, 0], (g * 0.5).astype(np.float32))
        assert np.array_equal(table[:, 5], (g * 0.5 + 5).astype(np.float32))
        assert np.array_equal(table[:, 6], (g * 0.25).astype(np.float32))
        assert np.array_equal(table[:, 7], (32 + 4 * (g %% 2)).astype(np.float32))
        # the preallocated gatherer bench.py uses (reused buffers over several batches, enqueue/fetch pipelining)
        gat = gloo_gather.ResultGatherer(total)
        for rep in range(3):
            pose, iters, wgt = FakeCtx(lo).align(slots, slots)
            t2 = gat.gather(sharding.pack_results(pose + rep, iters, wgt))
            assert t2.shape == (total, 8) and np.array_equal(t2[:, 0], (g * 0.5 + rep).astype(np.float32))
            assert np.array_equal(t2[:, 7], (32 + 4 * (g %% 2)).astype(np.float32))
        # deferred form: up to three gathers outstanding, collected oldest first (bench.py with three batches in flight)
        ring = gloo_gather.ResultGatherer(total, depth=3)
        for rep in range(7):
            if len(ring.pending) == 3:
                t3 = ring.finish()
                assert np.array_equal(t3[:, 0], (g * 0.5 + (rep - 3)).astype(np.float32))
            pose, iters, wgt = FakeCtx(lo).align(slots, slots)
            ring.start(sharding.pack_results(pose + rep, iters, wgt))
        for rep in (4, 5, 6):
            t3 = ring.finish().copy()
            assert t3.shape == (total, 8) and np.array_equal(t3[:, 0], (g * 0.5 + rep).astype(np.float32))
        try:
            ring.finish()
            raise SystemExit("finish() with nothing outstanding must fail")
        except RuntimeError:
            pass
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_shard_range_partitions():
    from egomotion_with_local_loop_closures_amd import sharding
    for total in (0, 1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = sharding.shard_range(total, world, r)
                assert 0 <= lo <= hi <= total
                seen += list(range(lo, hi))
            assert seen == list(range(total))
    assert sharding.shard_range(256, 8, 3) == (96, 128)     # C3: 32 per GPU, contiguous blocks


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "ok" in o


def test_gatherer_without_a_process_group_passes_tables_through():
    import gloo_gather
    gat = gloo_gather.ResultGatherer(3, depth=2)
    a = np.arange(24, dtype=np.float32).reshape(3, 8)
    gat.start(a)
    gat.start(a + 1)
    with pytest.raises(RuntimeError):
        gat.start(a)
    assert np.array_equal(gat.finish(), a) and np.array_equal(gat.finish(), a + 1)
    with pytest.raises(RuntimeError):
        gat.finish()
    assert np.array_equal(gat.gather(a + 2), a + 2)


WORKER_COMM = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    from egomotion_with_local_loop_closures_amd import sharding

    class FakeCtx:
        def __init__(self, lo): self.lo = lo
        def align(self, kf, fr, init_pose=None, mode=0):
            g = self.lo + np.arange(len(kf))
            pose = np.stack([g * 0.5 + k for k in range(6)], axis=1).astype(np.float32)
            iters = np.tile(np.array([[4, 7, 9, 12]]), (len(kf), 1)) + (g[:, None] %% 2)
            return pose, iters, (g * 0.25).astype(np.float32)

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    comm = sharding.Comm(world, rank, max_total=16, transport="tcp", port=int(os.environ["MASTER_PORT"]), host_only=True)
    for total in (8, 7, 1, 5):
        lo, hi = comm.shard_range(total)
        assert (lo, hi) == sharding.shard_range(total, world, rank)
        slots = np.arange(hi - lo)
        table = sharding.align_sharded(FakeCtx(lo), comm, total, slots, slots)
        g = np.arange(total)
        assert table.shape == (total, 8), table.shape
        assert np.array_equal(table[:, 0], (g * 0.5).astype(np.float32)) and np.array_equal(table[:, 5], (g * 0.5 + 5).astype(np.float32))
        assert np.array_equal(table[:, 6], (g * 0.25).astype(np.float32)) and np.array_equal(table[:, 7], (32 + 4 * (g %% 2)).astype(np.float32))
    comm.close()
    print("rank", rank, "ok")
""")


def test_two_rank_gather_through_the_library(tmp_path):
    """The product's path: sharding.align_sharded over sharding.Comm (ellc_gather_results of the C ABI, TCP transport, no GPU)."""
    script = tmp_path / "worker_comm.py"
    script.write_text(WORKER_COMM % ROOT)
    env = dict(os.environ, MASTER_PORT=str(29800 + os.getpid() % 100), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "ok" in o
