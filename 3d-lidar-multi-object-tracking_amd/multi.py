"""Multi-GPU harness pieces shared by bench.py and the 2-process gloo test.

The path shards by sensor stream: rank r owns streams r*B .. r*B+B-1 and its own context; there is no
collective inside the data path. The only exchange is the per-step all-gather of the fixed-size block of live-track
records of every stream (B x max_tracks x 144 bytes per rank), over RCCL on GPUs, gloo in the CPU test."""
from __future__ import annotations

TRACK_RECORD_BYTES = 144
TRACK_RECORD_WORDS = TRACK_RECORD_BYTES // 4


def scene_of(rank: int, slot: int, scenes_per_rank: int = 8) -> int:
    """synthetic scene id of a stream: distinct per rank, tiled over the rank's slots"""
    return 100 * rank + (slot % scenes_per_rank)


class TrackGather:
    """all-gather of the live-track blocks. `device` = "cuda" (RCCL) or "cpu" (gloo, emulated kernels).

    One flat int32 buffer per rank — the B x max_tracks records followed by the B counts — so that a step is ONE collective."""

    def __init__(self, batch: int, max_tracks: int, world: int, device: str, group=None):
        import torch
        self.torch = torch
        self.batch, self.max_tracks, self.world = batch, max_tracks, world
        self.group = group   # process group of the collective (None = the default group)
        nrec = batch * max_tracks * TRACK_RECORD_WORDS
        self.flat = torch.zeros(nrec + batch, dtype=torch.int32, device=device)
        self.src = self.flat[:nrec].view(batch, max_tracks, TRACK_RECORD_WORDS)
        self.cnt = self.flat[nrec:]
        self._nrec = nrec
        self.recv = None          # [world x (records + counts)]: ONE receive buffer, filled by all_gather_into_tensor
        self._point_views(torch.zeros(world * self.flat.numel(), dtype=torch.int32, device=device) if world > 1 else None)
        self._ext = {}   # context -> torch view of its HIP stream

    def _point_views(self, recv):
        """dst / dst_cnt: per-rank views of the receive buffer (with one rank and no collective: of the send buffer itself)"""
        L = self.flat.numel()
        self.recv = recv
        self.dst_flat = [recv[r * L:(r + 1) * L] for r in range(self.world)] if recv is not None else [self.flat]
        self.dst = [d[:self._nrec].view(self.batch, self.max_tracks, TRACK_RECORD_WORDS) for d in self.dst_flat]
        self.dst_cnt = [d[self._nrec:] for d in self.dst_flat]

    def step(self, ctx, force_collective: bool = False):
        """export this rank's block (async on the context stream), then exchange.

        On GPUs nothing here blocks the host: the context's HIP stream is wrapped as a torch ExternalStream, the export
        waits for the previous step's collective (which still reads the block), the collective waits for the export —
        stream ordering only, so the next step's kernels are already queued behind it. On the CPU (gloo test) the
        emulated library is synchronous anyway."""
        self.export(ctx)
        return self.exchange(force_collective)

    def export(self, ctx):
        torch = self.torch
        if self.src.is_cuda:
            ext = self._ext.get(id(ctx))
            if ext is None:
                ext = self._ext[id(ctx)] = torch.cuda.ExternalStream(ctx.lib.mot_stream(ctx._h))
            cur = torch.cuda.current_stream()
            ext.wait_stream(cur)      # the previous collective has consumed self.src / self.cnt
        ctx.export_tracks_dev(self.batch, self.src.data_ptr(), self.max_tracks, self.cnt.data_ptr())
        if self.src.is_cuda:
            cur.wait_stream(ext)      # the block is complete before the collective reads it
        else:
            ctx.synchronize()

    def exchange(self, force_collective: bool = False):
        if self.world > 1 or force_collective:
            import torch.distributed as dist
            if self.recv is None:   # one rank, collective forced (bench.py --force-gather, tools/check_gather_gpu.py): its own receive buffer
                self._point_views(self.torch.zeros_like(self.flat))
            # one contiguous receive buffer: no list of per-rank tensors for c10d to flatten and copy back, one enqueue per step
            dist.all_gather_into_tensor(self.recv, self.flat, group=self.group)
        return self.dst, self.dst_cnt

    def blocks_as_numpy(self):
        """[(counts[B], records[B][max_tracks] structured)] per rank — host copy, for tests"""
        import numpy as np
        rec = np.dtype([("id", "i4"), ("track_manage", "i4"), ("is_static", "i4"), ("is_vis", "i4"), ("p", "f4", 3),
                        ("lifetime", "i4"), ("v_yaw", "f8", 2), ("vis_box", "f4", 24)])
        out = []
        for d, c in zip(self.dst, self.dst_cnt):
            a = d.cpu().numpy().view(np.uint8).reshape(self.batch, self.max_tracks, TRACK_RECORD_BYTES)
            out.append((c.cpu().numpy().copy(), a.view(rec).reshape(self.batch, self.max_tracks).copy()))
        return out


def packed_block_bytes(batch: int, capacity_records: int) -> int:
    """size of one context's packed block (include/mot.h, mot_export_tracks_packed_dev): counts header padded to 16 bytes + records"""
    return ((batch * 4 + 15) & ~15) + capacity_records * TRACK_RECORD_BYTES


class TrackGatherAll:
    """ONE all-gather per frame for ALL contexts of a rank, of PACKED blocks (mot_export_tracks_packed_dev: per context a header of
    per-stream counts + the live records back to back) — what round 2's per-context fixed-slot gather cost in bytes (64 slots x
    144 B per stream whatever is alive: 3.8x padding at 17 live tracks) and in collectives (one communicator and one enqueue per
    context and frame) is what this class removes.

    Two send / receive buffer pairs alternate, so a context exports frame t+1 while the collective of frame t is in flight; all
    ordering is by stream / event waits (torch ExternalStream views of the contexts' HIP streams), the host never blocks. Every
    rank calls step() the same number of times, in the same order: one process group, one collective per step."""

    def __init__(self, ctxs, batch: int, capacity_records: int, world: int, device: str, group=None):
        import torch
        self.torch = torch
        self.ctxs, self.batch, self.cap, self.world, self.group = list(ctxs), batch, capacity_records, world, group
        self.nc = len(self.ctxs)
        self.block = packed_block_bytes(batch, capacity_records)
        self.head = (batch * 4 + 15) & ~15
        self.cuda = device != "cpu"
        mk = lambda n: torch.zeros(n, dtype=torch.uint8, device=device)
        self.send = [mk(self.nc * self.block) for _ in range(2)]
        self.recv = [mk(world * self.nc * self.block) for _ in range(2)]
        self.tick = 0
        if self.cuda:
            self.side = torch.cuda.Stream()
            self.ext = [torch.cuda.ExternalStream(cx.lib.mot_stream(cx._h)) for cx in self.ctxs]
            self.exported = [torch.cuda.Event() for _ in self.ctxs]
            self.done = [None, None]   # the collective that last used buffer pair i

    def step(self, force_collective: bool = False):
        """export every context's packed block (async, each on its context's stream), then ONE collective for all of them.
        Returns the receive buffer of this step: complete once the side stream has run (synchronize(), or a wait on `self.done[self.last]`
        from the consumer's stream), and REWRITTEN by the step after next — a consumer that needs it longer copies it."""
        torch = self.torch
        i = self.tick & 1
        self.tick += 1
        send = self.send[i]
        for ci, cx in enumerate(self.ctxs):
            if self.cuda and self.done[i] is not None:
                self.ext[ci].wait_event(self.done[i])          # the collective two steps ago has read this buffer
            cx.export_tracks_packed_dev(self.batch, send.data_ptr() + ci * self.block, self.block)
            if self.cuda:
                self.exported[ci].record(self.ext[ci])
                self.side.wait_event(self.exported[ci])
            else:
                cx.synchronize()
        if self.world > 1 or force_collective:
            import torch.distributed as dist
            if self.cuda:
                with torch.cuda.stream(self.side):
                    dist.all_gather_into_tensor(self.recv[i], send, group=self.group)
                    ev = torch.cuda.Event(); ev.record(self.side); self.done[i] = ev
            else:
                dist.all_gather_into_tensor(self.recv[i], send, group=self.group)
        elif self.cuda:   # one rank, no collective: the block itself — on the side stream, which waits for the exports (like the collective)
            with torch.cuda.stream(self.side):
                self.recv[i][: send.numel()].copy_(send, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(self.side); self.done[i] = ev
        else:
            self.recv[i][: send.numel()] = send
        self.last = i
        return self.recv[i]

    def synchronize(self):
        if self.cuda:
            self.side.synchronize()

    def blocks_as_numpy(self):
        """[rank][context] -> (counts[B], [records of stream 0, records of stream 1, ...], truncated) of the LAST step — host copy, for tests"""
        import numpy as np
        rec = np.dtype([("id", "i4"), ("track_manage", "i4"), ("is_static", "i4"), ("is_vis", "i4"), ("p", "f4", 3),
                        ("lifetime", "i4"), ("v_yaw", "f8", 2), ("vis_box", "f4", 24)])
        self.synchronize()
        raw = self.recv[self.last].cpu().numpy()
        out = []
        for r in range(self.world):
            per = []
            for ci in range(self.nc):
                blk = raw[(r * self.nc + ci) * self.block:(r * self.nc + ci + 1) * self.block]
                counts = blk[: self.batch * 4].view(np.int32).copy()
                total = int(counts.sum())
                recs = blk[self.head: self.head + min(total, self.cap) * TRACK_RECORD_BYTES].view(rec).copy()
                off = np.concatenate([[0], np.cumsum(counts)])
                per.append((counts, [recs[off[b]: min(off[b + 1], len(recs))] for b in range(self.batch)], total > self.cap))
            out.append(per)
        return out
