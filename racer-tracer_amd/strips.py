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
    def __init__(self, height, width, strip_rows, world, rank, device, dist=None, always_collective=False):
        """always_collective: go through dist.gather even when world == 1 (a one-rank process group: what
        `bench.py --force-dist` uses to prove on a one-GPU box that RCCL loads, initialises and gathers f64
        device tensors before a multi-GPU node meets this code)."""
        self.h, self.w, self.rows = height, width, strip_rows
        self.world, self.rank, self.dist = world, rank, dist
        self.collective = world > 1 or (always_collective and dist is not None)
        self.n_strips = (height + strip_rows - 1) // strip_rows
        self.per_rank = (self.n_strips + world - 1) // world  # padded so every rank sends the same size
        self.pad_rows = self.per_rank * world * strip_rows    # a whole number of strips for every rank
        self.padded = None
        if self.collective:
            kw = dict(dtype=torch.float64, device=device)
            # strip j = slot (j // world, j % world): one strided view per rank, one permute for all of them
            self.padded = torch.zeros((self.pad_rows, width, 3), **kw)
            self.by_rank = self.padded.view(self.per_rank, world, strip_rows, width, 3)
            self.send = torch.zeros((self.per_rank, strip_rows, width, 3), **kw)
            self.recv = torch.zeros((world, self.per_rank, strip_rows, width, 3), **kw) if rank == 0 else None

    def frame(self):
        """An [H, W, 3] frame to render into whose storage IS the gather's staging buffer, so
        gather() moves nothing but the strips themselves.  (Any other [H, W, 3] tensor works
        too, at the price of one copy in and one out.)"""
        if self.padded is None:
            raise RuntimeError("StripGather.frame() is only meaningful when a collective runs (world > 1)")
        return self.padded[: self.h]

    def owned(self, rank=None):
        """Strip indices owned by `rank` (default: this rank)."""
        return range(self.rank if rank is None else rank, self.n_strips, self.world)

    def owned_row_mask(self, rank=None):
        rows = torch.arange(self.h)
        r = self.rank if rank is None else rank
        return ((rows // self.rows) % self.world) == r

    def gather(self, frame):
        """frame: [H, W, 3] float64 holding this rank's rows.  After the call
        rank 0's frame holds every row.  No-op for world == 1 (unless always_collective).
        Per step: one strided copy into the send buffer, ONE collective, and on rank 0 one
        permuting copy back into the frame."""
        if not self.collective:
            return frame
        own_storage = frame.data_ptr() == self.padded.data_ptr()
        if not own_storage:
            self.padded[: self.h].copy_(frame)
        self.send.copy_(self.by_rank[:, self.rank])
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
            self.dist.gather(self.send, [self.recv[r] for r in range(self.world)] if self.rank == 0 else None, dst=0)
        if self.rank == 0:
            self.by_rank.copy_(self.recv.permute(1, 0, 2, 3, 4))
            if not own_storage:
                frame.copy_(self.padded[: self.h])
        return frame
