"""Row-strip ownership and the gather of finished strips to rank 0.

The reference shards a frame into tiles for its rayon pool
(racer-tracer/src/renderer/cpu.rs:73-131).  Across GPUs the same idea is
applied to rows: the frame is cut into strips of `strip_rows` rows and strip j
belongs to rank j % world (interleaved, so cheap sky rows and expensive floor
rows spread evenly).  Each rank renders only its strips (RtRenderParams
strip_*), then ONE gather moves them to rank 0 — RCCL over xGMI on GPUs
(backend "nccl"), gloo in the CPU tests.  There is no other collective on the
path; the RNG is keyed by the global pixel index, so the assembled frame is
bit-identical for every world size.
"""
import torch


class StripGather:
    def __init__(self, height, width, strip_rows, world, rank, device, dist=None):
        self.h, self.w, self.rows = height, width, strip_rows
        self.world, self.rank, self.dist = world, rank, dist
        self.n_strips = (height + strip_rows - 1) // strip_rows
        self.per_rank = (self.n_strips + world - 1) // world  # padded so every rank sends the same size
        self.pad_rows = self.n_strips * strip_rows
        if world > 1:
            kw = dict(dtype=torch.float64, device=device)
            self.padded = torch.zeros((self.pad_rows, width, 3), **kw)
            self.send = torch.zeros((self.per_rank, strip_rows, width, 3), **kw)
            self.recv = [torch.zeros_like(self.send) for _ in range(world)] if rank == 0 else None

    def owned(self, rank=None):
        """Strip indices owned by `rank` (default: this rank)."""
        return range(self.rank if rank is None else rank, self.n_strips, self.world)

    def owned_row_mask(self, rank=None):
        rows = torch.arange(self.h)
        r = self.rank if rank is None else rank
        return ((rows // self.rows) % self.world) == r

    def gather(self, frame):
        """frame: [H, W, 3] float64 holding this rank's rows.  After the call
        rank 0's frame holds every row.  No-op for world == 1."""
        if self.world == 1:
            return frame
        self.padded[: self.h].copy_(frame)
        strips = self.padded.view(self.n_strips, self.rows, self.w, 3)
        mine = strips[self.rank :: self.world]
        self.send[: mine.shape[0]].copy_(mine)
        if self.send.is_cuda and self.dist.get_backend() == "gloo":
            # functional rehearsal of the N > 1 path on one GPU (several ranks share the
            # card, RCCL cannot): stage the collective through host memory
            send = self.send.cpu()
            recv = [torch.zeros_like(send) for _ in range(self.world)] if self.rank == 0 else None
            self.dist.gather(send, recv, dst=0)
            if self.rank == 0:
                for r in range(self.world):
                    self.recv[r].copy_(recv[r])
        else:
            self.dist.gather(self.send, self.recv, dst=0)
        if self.rank == 0:
            for r in range(self.world):
                k = len(self.owned(r))
                strips[r :: self.world] = self.recv[r][:k]
            frame.copy_(self.padded[: self.h])
        return frame
