"""The N > 1 path on CPU: two `gloo` ranks each render their row strips (with
the oracle standing in for the GPU — same RtRenderParams.strip_* contract),
gather them with racer-tracer_amd/strips.py exactly as bench.py does, and rank
0 must end up with the single-rank frame bit for bit."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, spp, rows, in_place, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import scenes_py as S
    from oracle import oracle_ctypes as orc
    strips = importlib.import_module("racer-tracer_amd.strips")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bundle, cam, _ = S.three_balls()
        camera = S.camera_for(cam, w, h)
        params = S.abi.render_params(w, h, spp, strip_rows=rows, strip_count=world, strip_index=rank)
        part, _ = orc.render(bundle.desc, camera, params, n_threads=2)
        g = strips.StripGather(h, w, rows, world, rank, "cpu", dist)
        if in_place:   # bench.py's way: the frame is the gather's own staging buffer
            frame = g.frame()
            frame.copy_(torch.from_numpy(part))
        else:
            frame = torch.from_numpy(part.copy())
        assert list(g.owned()) == list(range(rank, g.n_strips, world))
        assert bool((frame[~g.owned_row_mask()] == 0).all())
        g.gather(frame)
        dist.barrier()
        if rank == 0:
            np.save(out_path, frame.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,h,rows,in_place", [(2, 45, 8, True), (2, 40, 4, False), (3, 50, 8, True), (3, 22, 8, False)])
def test_two_rank_gather_equals_single_rank_frame(tmp_path, orc, world, h, rows, in_place):
    import scenes_py as S
    w, spp = 48, 3
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, spp, rows, in_place, out), nprocs=world, join=True)
    got = np.load(out)
    bundle, cam, _ = S.three_balls()
    full, _ = orc.render(bundle.desc, S.camera_for(cam, w, h), S.abi.render_params(w, h, spp))
    assert np.array_equal(got, full)


def _pattern(h, w):
    """A frame whose every element names its own (row, column, channel)."""
    r = torch.arange(h, dtype=torch.float64).view(h, 1, 1)
    c = torch.arange(w, dtype=torch.float64).view(1, w, 1)
    k = torch.arange(3, dtype=torch.float64).view(1, 1, 3)
    return r * 4096.0 + c + k * 0.25


def _pattern_worker(rank, world, port, w, h, rows, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    strips = importlib.import_module("racer-tracer_amd.strips")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = strips.StripGather(h, w, rows, world, rank, "cpu", dist)
        frame = g.frame()                       # bench.py's way: render into the gather's own staging buffer
        own = g.owned_row_mask()
        frame[own] = _pattern(h, w)[own]        # this rank's strips only; everything else stays zero
        assert int(own.sum()) == sum(min(rows, h - j * rows) for j in g.owned())
        g.gather(frame)
        dist.barrier()
        if rank == 0:
            np.save(out_path, frame.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,rows", [(8, 1920, 1080, 8),    # the bench frame: 135 strips over 8 ranks, 17 slots each, one padded
                                            (8, 64, 2160, 8),      # BASELINE config 5's rows: 270 strips, 34 slots, two padded
                                            (4, 1920, 1080, 8),    # 135 over 4: three ranks own 34 strips, one owns 33
                                            (8, 48, 1083, 8)])     # a short last strip (3 rows) on top of the padding
def test_gather_padding_at_real_frame_sizes(tmp_path, world, w, h, rows):
    """StripGather pads every rank to the same number of strip slots when n_strips % world != 0.  The render tests above
    cover it at heights of 22-50 rows; this is the gather ALONE at the sizes bench.py --gpus 8 and config 5 run it
    (cpu.rs:118-131 collects tiles of the real frame), with a frame whose every element names its own position, so a
    strip that lands in the wrong slot, a padded slot that leaks into the frame or a truncated last strip shows."""
    out = str(tmp_path / "frame.npy")
    mp.spawn(_pattern_worker, args=(world, _free_port(), w, h, rows, out), nprocs=world, join=True)
    assert np.array_equal(np.load(out), _pattern(h, w).numpy())


def test_strip_gather_single_rank_is_a_no_op():
    strips = importlib.import_module("racer-tracer_amd.strips")
    g = strips.StripGather(20, 6, 8, 1, 0, "cpu")
    f = torch.rand((20, 6, 3), dtype=torch.float64)
    assert g.gather(f) is f and g.n_strips == 3 and list(g.owned()) == [0, 1, 2]
    assert bool(g.owned_row_mask().all())
