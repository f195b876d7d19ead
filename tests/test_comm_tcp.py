"""The C ABI's multi-GPU layer on CPU: two (and three) processes drive ellc_gather_start / _finish / _results of
csrc/libellc_comm.so — the same C++ entry points libellc_hip.so exports, here over the TCP transport — on shards of uneven
size, with several gathers outstanding, and check order and padding of the gathered table."""
import os
import subprocess
import sys
import textwrap
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    import numpy as np
    from egomotion_with_local_loop_closures_amd import sharding, _lib
    world, rank, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    comm = sharding.Comm(world, rank, max_total=40, transport="tcp", port=port, host_only=True)
    # what the transport itself saw (ellc_comm_info: the self-check bench.py carries in its N > 1 line; RCCL reports ncclCommCount /
    # ncclCommUserRank / the device's PCI bus id through the same call)
    info = comm.info()
    assert info == {"transport": "tcp", "world_seen": world, "rank_seen": rank, "pci_bus_id": ""}, info

    def shard(total, salt):
        lo, hi = comm.shard_range(total)
        assert (lo, hi) == sharding.shard_range(total, world, rank)          # the C and the Python partition agree
        g = np.arange(lo, hi)
        t = np.zeros((hi - lo, 8), np.float32)
        for k in range(6):
            t[:, k] = g * 0.5 + k + salt
        t[:, 6] = g * 0.25
        t[:, 7] = 32 + 4 * (g %% 2)
        return t

    def expect(total, salt):
        g = np.arange(total)
        t = np.zeros((total, 8), np.float32)
        for k in range(6):
            t[:, k] = g * 0.5 + k + salt
        t[:, 6] = g * 0.25
        t[:, 7] = 32 + 4 * (g %% 2)
        return t

    for total in (8, 7, 1, 5, 40, 33):
        assert np.array_equal(comm.gather(total, shard(total, 0)), expect(total, 0)), total
    # several gathers outstanding, collected oldest first (a pipeline of batches)
    for rep in range(4):
        comm.start(9, shard(9, rep))
    for rep in range(4):
        assert np.array_equal(comm.finish(9), expect(9, rep))
    # outstanding gathers of DIFFERENT sizes: finish() returns the oldest, sized by its own total (r02 sized it by the newest: a
    # heap overflow); a wrong explicit total is refused and the gather stays outstanding; the C side refuses a short buffer
    comm.start(40, shard(40, 7)); comm.start(8, shard(8, 8)); comm.start(33, shard(33, 9))
    try:
        comm.finish(8)
        raise SystemExit("finish(8) accepted although the oldest gather has 40 records")
    except _lib.EllcError:
        pass
    import ctypes as C
    small = np.zeros((8, 8), np.float32)
    assert comm._l.ellc_gather_finish(comm.h, small.ctypes.data_as(C.c_void_p), 8) == -5     # ELLC_ERR_CAPACITY, nothing written
    assert not small.any()
    assert np.array_equal(comm.finish(), expect(40, 7))
    assert np.array_equal(comm.finish(8), expect(8, 8))
    assert np.array_equal(comm.finish(33), expect(33, 9))
    # misuse is refused
    for bad in (lambda: comm.finish(9), lambda: comm.start(41, shard(40, 0)), lambda: comm.start(8, np.zeros((0 if rank else 1, 8), np.float32))):
        try:
            bad()
            raise SystemExit("a bad call was accepted")
        except _lib.EllcError:
            pass
    comm.close()
    print("rank", rank, "ok")
""") % ROOT


@pytest.mark.parametrize("world", [2, 3])
def test_gather_entry_points_over_tcp(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29650 + world + (os.getpid() % 200)
    procs = [subprocess.Popen([sys.executable, str(script), str(world), str(r), str(port)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "ok" in o


def test_single_rank_is_a_copy():
    from egomotion_with_local_loop_closures_amd import sharding
    comm = sharding.Comm(1, 0, max_total=16, transport="tcp", host_only=True)
    assert comm.info()["world_seen"] == 1 and comm.info()["rank_seen"] == 0
    t = np.arange(40, dtype=np.float32).reshape(5, 8)
    assert np.array_equal(comm.gather(5, t), t)
    comm.close()
