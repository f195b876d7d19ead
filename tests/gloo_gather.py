"""The per-batch gather of the sharded alignments' results over torch.distributed (gloo on CPU) — TEST infrastructure: the
product gathers through the C ABI (sharding.Comm: RCCL / TCP); this restates the same exchange with torch's collectives so that the
partition, the padding of uneven shards and the order of the gathered table are also checked against an independent transport
(tests/test_sharding_gloo.py, world size 2)."""
import numpy as np
from egomotion_with_local_loop_closures_amd.sharding import RECORD, pack_results


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


def align_sharded_gloo(ctx, total, local_kf_slots, local_frame_slots, init_pose=None, mode=0, device=None, group=None):
    """this rank's share of `total` alignments, then the torch.distributed gather"""
    pose, iters, wgt = ctx.align(local_kf_slots, local_frame_slots, init_pose=init_pose, mode=mode)
    return gather_results(pack_results(pose, iters, wgt), total, device=device, group=group)
