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


def test_strip_gather_single_rank_is_a_no_op():
    strips = importlib.import_module("racer-tracer_amd.strips")
    g = strips.StripGather(20, 6, 8, 1, 0, "cpu")
    f = torch.rand((20, 6, 3), dtype=torch.float64)
    assert g.gather(f) is f and g.n_strips == 3 and list(g.owned()) == [0, 1, 2]
    assert bool(g.owned_row_mask().all())
