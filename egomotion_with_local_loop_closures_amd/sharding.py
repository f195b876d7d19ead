"""Multi-GPU layer: independent keyframe<->frame alignments are sharded across ranks (one process per GPU);
the only data-path collective is ONE gather of the resulting se(3) poses per batch.

The reference has no distributed code; alignment b of a loop-closure batch (GlobalOptimize.cpp:566) is
independent of every other, so ranks never exchange pixels — only 8 floats per alignment at the end:
[pose(6), weightedPose, iterations]. The exchange is the library's own (Comm: ellc_gather_* of the C ABI — ncclAllGather over
xGMI, or the TCP transport); nothing here imports torch. (The torch.distributed form of the same gather that the world-size-2
gloo test runs lives with that test: tests/gloo_gather.py.)
"""
import numpy as np

RECORD = 8  # floats per alignment in the gathered table


def shard_range(total, world, rank):
    """Contiguous block partition: alignment b lives on rank b // ceil(total / world)."""
    per = (total + world - 1) // world
    lo = min(total, rank * per)
    hi = min(total, lo + per)
    return lo, hi


def pack_results(pose, iters, weighted):
    pose = np.asarray(pose, np.float32).reshape(-1, 6)
    out = np.zeros((pose.shape[0], RECORD), np.float32)
    out[:, :6] = pose
    out[:, 6] = np.asarray(weighted, np.float32).reshape(-1)
    if pose.shape[0]:
        out[:, 7] = np.asarray(iters).reshape(pose.shape[0], -1).sum(axis=1)
    return out


class Comm:
    """The C ABI's communicator (include/ellc_abi.h, multi-GPU section): the gather of a sharded batch's results as the
    library implements it in C++ — ncclAllGather over xGMI (transport "rccl", libellc_hip.so) or a TCP star through rank 0
    (transport "tcp", host memory only; `host_only` loads csrc/libellc_comm.so, which needs no GPU). torch.distributed is
    not involved; a launcher only has to hand rank 0's 128-byte unique id to the other ranks (rccl) or name a port (tcp)."""

    def __init__(self, world, rank, max_total, transport="tcp", device=0, unique_id=None, host="127.0.0.1", port=29611, host_only=False):
        import ctypes as C
        from . import _lib
        self._C = C
        self._l = _lib.comm_lib() if host_only else _lib.lib()
        self.world, self.rank, self.max_total = world, rank, max_total
        self._totals = []   # totals of the outstanding gathers, oldest first (ellc_gather_finish returns the oldest)
        h = C.c_void_p()
        if transport == "rccl":
            assert unique_id is not None and len(unique_id) == 128
            buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
            st = self._l.ellc_comm_init_rccl(device, buf, world, rank, max_total, C.byref(h))
        else:
            st = self._l.ellc_comm_init_tcp(host.encode(), port, world, rank, max_total, C.byref(h))
        self.h = h
        if st != 0:
            msg = self._l.ellc_comm_last_error(h).decode() if h else "communicator creation failed"
            self.close()
            raise _lib.EllcError("ellc_comm_init_%s -> %d: %s" % (transport, st, msg))

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib
        buf = (C.c_ubyte * 128)()
        st = _lib.lib().ellc_comm_unique_id(buf)
        if st != 0:
            raise _lib.EllcError("ellc_comm_unique_id -> %d" % st)
        return bytes(buf)

    def info(self):
        """What the transport itself reports (ellc_comm_info): {'transport', 'world_seen', 'rank_seen', 'pci_bus_id'}."""
        C = self._C
        t, w, r = C.c_int(0), C.c_int(0), C.c_int(-1)
        bus = C.create_string_buffer(64)
        st = self._l.ellc_comm_info(self.h, C.byref(t), C.byref(w), C.byref(r), bus, 64)
        if st != 0:
            raise RuntimeError("ellc_comm_info -> %d" % st)
        return {"transport": {1: "rccl", 2: "tcp"}.get(t.value, str(t.value)), "world_seen": w.value, "rank_seen": r.value, "pci_bus_id": bus.value.decode()}

    def shard_range(self, total):
        C = self._C
        lo, hi = C.c_int(0), C.c_int(0)
        self._l.ellc_shard_range(total, self.world, self.rank, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def _ck(self, st, what):
        if st != 0:
            from . import _lib
            raise _lib.EllcError("%s -> %d: %s" % (what, st, self._l.ellc_comm_last_error(self.h).decode()))

    def start(self, total, local):
        loc = np.ascontiguousarray(local, np.float32).reshape(-1, RECORD)
        self._ck(self._l.ellc_gather_start(self.h, total, loc.ctypes.data_as(self._C.c_void_p), loc.shape[0]), "ellc_gather_start")
        self._totals.append(total)

    def finish(self, total=None):
        """The table of the OLDEST outstanding gather. `total`, when given, must be that gather's total."""
        if not self._totals:
            from . import _lib
            raise _lib.EllcError("Comm.finish: no gather outstanding")
        if total is not None and total != self._totals[0]:
            from . import _lib
            raise _lib.EllcError("Comm.finish: the oldest outstanding gather has %d records, not %d" % (self._totals[0], total))
        out = np.zeros((self._totals[0], RECORD), np.float32)
        self._ck(self._l.ellc_gather_finish(self.h, out.ctypes.data_as(self._C.c_void_p), out.shape[0]), "ellc_gather_finish")
        self._totals.pop(0)
        return out

    def gather(self, total, local):
        loc = np.ascontiguousarray(local, np.float32).reshape(-1, RECORD)
        out = np.zeros((total, RECORD), np.float32)
        self._ck(self._l.ellc_gather_results(self.h, total, loc.ctypes.data_as(self._C.c_void_p), loc.shape[0], out.ctypes.data_as(self._C.c_void_p)),
                 "ellc_gather_results")
        return out

    def close(self):
        if getattr(self, "h", None):
            self._l.ellc_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def align_sharded(ctx, comm, total, local_kf_slots, local_frame_slots, init_pose=None, mode=0):
    """Run this rank's share of `total` alignments on its GPU and gather all poses through `comm` (a Comm: the library's gather).

    local_*_slots index THIS rank's resident slots, in the order of the rank's global range."""
    pose, iters, wgt = ctx.align(local_kf_slots, local_frame_slots, init_pose=init_pose, mode=mode)
    return comm.gather(total, pack_results(pose, iters, wgt))
