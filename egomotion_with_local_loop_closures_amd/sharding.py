"""Multi-GPU layer: independent keyframe<->frame alignments are sharded across ranks (one process per GPU);
the only data-path collective is ONE gather of the resulting se(3) poses per batch.

The reference has no distributed code; alignment b of a loop-closure batch (GlobalOptimize.cpp:566) is
independent of every other, so ranks never exchange pixels — only 8 floats per alignment at the end:
[pose(6), weightedPose, iterations]. Backend "nccl" (= RCCL over xGMI on MI355X) on GPUs, "gloo" in CPU tests.
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


def gather_results(local, total, device=None, group=None):
    """All ranks contribute their (n_local, 8) table; every rank gets the (total, 8) table in global order.

    One all_gather of fixed-size (padded) blocks: 32 alignments x 32 B = 1 KiB per rank — latency-bound, so the
    choice of ring vs direct and the xGMI link budget are irrelevant.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return np.asarray(local, np.float32).reshape(-1, RECORD)[:total]
    world = dist.get_world_size(group)
    per = (total + world - 1) // world
    buf = torch.zeros((per, RECORD), dtype=torch.float32)
    loc = torch.from_numpy(np.ascontiguousarray(local, np.float32).reshape(-1, RECORD))
    buf[: loc.shape[0]] = loc
    if device is not None:
        buf = buf.to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    table = torch.cat(parts, dim=0)[:total]
    return table.cpu().numpy()


class ResultGatherer:
    """The per-batch gather with everything allocated once. A ring of `depth` buffer sets — a (per, 8) input block and a
    (world*per, 8) output table on the collective's device (GPU for nccl = RCCL, CPU for gloo), pinned host copies of both
    and an event — lets up to `depth` gathers be outstanding: start() only enqueues (small H2D, all_gather_into_tensor,
    D2H, event) on torch's stream, finish() waits for the OLDEST outstanding gather and returns its table. A caller that
    keeps several batches in flight on the library's streams starts the gather of batch s when it fetches it and finishes
    it a few steps later, so neither the exchange nor a collective kernel that queues behind a batch on a shared hardware
    queue ever stalls the host loop. gather() = start() + finish() for one-at-a-time use."""

    def __init__(self, total, device=None, group=None, depth=1):
        import torch
        import torch.distributed as dist
        self.total, self.group = total, group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.depth = max(1, int(depth))
        self.pending = []          # ring indices (active) or tables (inactive) of the outstanding gathers, oldest first
        self.cursor = 0
        if not self.active:
            return
        self.world = dist.get_world_size(group)
        self.per = (total + self.world - 1) // self.world
        self.on_gpu = device is not None and torch.device(device).type == "cuda"
        dev = torch.device(device) if device is not None else torch.device("cpu")
        self.into_tensor = dist.get_backend(group) != "gloo"   # gloo: list form
        self.ring = []
        for _ in range(self.depth):
            host_in = torch.zeros((self.per, RECORD), dtype=torch.float32)
            host_out = torch.zeros((self.world * self.per, RECORD), dtype=torch.float32)
            if self.on_gpu:
                host_in, host_out = host_in.pin_memory(), host_out.pin_memory()
            self.ring.append({
                "host_in": host_in, "host_out": host_out,
                "dev_in": torch.zeros((self.per, RECORD), dtype=torch.float32, device=dev),
                "dev_out": torch.zeros((self.world * self.per, RECORD), dtype=torch.float32, device=dev),
                "event": torch.cuda.Event() if self.on_gpu else None})

    def start(self, local):
        """Enqueue the gather of this rank's (n_local, 8) table; at most `depth` may be outstanding."""
        if len(self.pending) >= self.depth:
            raise RuntimeError("ResultGatherer: %d gathers outstanding, finish() one first" % self.depth)
        loc = np.asarray(local, np.float32).reshape(-1, RECORD)
        if not self.active:
            self.pending.append(loc[: self.total].copy())
            return
        import torch
        import torch.distributed as dist
        r = self.ring[self.cursor]
        r["host_in"].zero_()
        r["host_in"][: loc.shape[0]] = torch.from_numpy(loc)
        r["dev_in"].copy_(r["host_in"], non_blocking=True)
        if self.into_tensor:
            dist.all_gather_into_tensor(r["dev_out"], r["dev_in"], group=self.group)
        else:
            parts = list(r["dev_out"].view(self.world, self.per, RECORD).unbind(0))
            dist.all_gather(parts, r["dev_in"], group=self.group)
        r["host_out"].copy_(r["dev_out"], non_blocking=True)
        if r["event"] is not None:
            r["event"].record()
        self.pending.append(self.cursor)
        self.cursor = (self.cursor + 1) % self.depth

    def finish(self):
        """Wait for the oldest outstanding gather; returns its (total, 8) table in global order (a view of a ring buffer,
        valid until `depth` further gathers have been started)."""
        if not self.pending:
            raise RuntimeError("ResultGatherer: no gather outstanding")
        head = self.pending.pop(0)
        if not self.active:
            return head
        r = self.ring[head]
        if r["event"] is not None:
            r["event"].synchronize()
        return r["host_out"].numpy()[: self.total]

    def gather(self, local):
        """local: (n_local, 8) table of this rank; returns the (total, 8) table in global order."""
        self.start(local)
        return self.finish()


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


def align_sharded(ctx, total, local_kf_slots, local_frame_slots, init_pose=None, mode=0, device=None, group=None):
    """Run this rank's share of `total` alignments on its GPU and gather all poses.

    local_*_slots index THIS rank's resident slots, in the order of the rank's global range."""
    pose, iters, wgt = ctx.align(local_kf_slots, local_frame_slots, init_pose=init_pose, mode=mode)
    return gather_results(pack_results(pose, iters, wgt), total, device=device, group=group)
