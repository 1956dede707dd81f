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
    """all-gather of the live-track blocks. `device` = "cuda" (RCCL) or "cpu" (gloo, emulated kernels)."""

    def __init__(self, batch: int, max_tracks: int, world: int, device: str):
        import torch
        self.torch = torch
        self.batch, self.max_tracks, self.world = batch, max_tracks, world
        self.src = torch.zeros(batch, max_tracks, TRACK_RECORD_WORDS, dtype=torch.int32, device=device)
        self.cnt = torch.zeros(batch, dtype=torch.int32, device=device)
        self.dst = [torch.zeros_like(self.src) for _ in range(world)] if world > 1 else [self.src]
        self.dst_cnt = [torch.zeros_like(self.cnt) for _ in range(world)] if world > 1 else [self.cnt]
        self._ext = {}   # context -> torch view of its HIP stream

    def step(self, ctx, force_collective: bool = False):
        """export this rank's block (async on the context stream), then exchange.

        On GPUs nothing here blocks the host: the context's HIP stream is wrapped as a torch ExternalStream, the export
        waits for the previous step's collective (which still reads the block), the collective waits for the export —
        stream ordering only, so the next step's kernels are already queued behind it. On the CPU (gloo test) the
        emulated library is synchronous anyway."""
        torch = self.torch
        gpu = self.src.is_cuda
        if gpu:
            ext = self._ext.get(id(ctx))
            if ext is None:
                ext = self._ext[id(ctx)] = torch.cuda.ExternalStream(ctx.lib.mot_stream(ctx._h))
            cur = torch.cuda.current_stream()
            ext.wait_stream(cur)      # the previous collective has consumed self.src / self.cnt
        ctx.export_tracks_dev(self.batch, self.src.data_ptr(), self.max_tracks, self.cnt.data_ptr())
        if gpu:
            cur.wait_stream(ext)      # the block is complete before the collective reads it
        else:
            ctx.synchronize()
        if self.world > 1 or force_collective:
            import torch.distributed as dist
            dist.all_gather(self.dst, self.src)
            dist.all_gather(self.dst_cnt, self.cnt)
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
