"""Whole-frame behaviour of the CPU oracle: determinism, BVH == linear scan,
strip ownership, thread-count independence, and the committed golden frames."""
import importlib
import os
import sys

import numpy as np
import pytest

import scenes_py as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_oracle_fixtures as fx  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "oracle_frames.npz"))


@pytest.fixture(scope="module")
def host():
    return importlib.import_module("racer-tracer_amd.host")


def test_deterministic_and_thread_count_independent(orc):
    bundle, cam, _ = S.three_balls()
    camera = S.camera_for(cam, 80, 45)
    p = S.abi.render_params(80, 45, 4)
    a, sa = orc.render(bundle.desc, camera, p, n_threads=1)
    b, sb = orc.render(bundle.desc, camera, p, n_threads=7)
    assert np.array_equal(a, b) and sa == sb
    c, _ = orc.render(bundle.desc, camera, S.abi.render_params(80, 45, 4, tiles_w=3, tiles_h=7))
    assert np.array_equal(a, c)          # tiling does not change pixels: the RNG is keyed by global pixel index
    d, _ = orc.render(bundle.desc, camera, S.abi.render_params(80, 45, 4, seed=2))
    assert not np.array_equal(a, d)


@pytest.mark.parametrize("scene_fn", [S.three_balls, S.cornell_box, S.two_balls])
def test_bvh_equals_linear_scan(orc, scene_fn):
    """Closest-hit semantics are topology-free (SURVEY B-15): the reference's
    BVH walk and a brute-force scan give the same image on untransformed scenes."""
    bundle, cam, _ = scene_fn()
    camera = S.camera_for(cam, 96, 54)
    p = S.abi.render_params(96, 54, 6)
    a, sa = orc.render(bundle.desc, camera, p, use_bvh=1)
    b, sb = orc.render(bundle.desc, camera, p, use_bvh=0)
    assert np.array_equal(a, b) and sa == sb


def test_rotate_y_bounding_box_bug_only_culls_a_sliver(orc):
    """With RotateY the reference's mis-sized AABB (rotate_y.rs:66-90) makes
    the BVH miss a thin strip of each box; everything else is identical."""
    bundle, cam, _ = S.cornell_box_boxes()
    camera = S.camera_for(cam, 160, 90)
    p = S.abi.render_params(160, 90, 4)
    a, _ = orc.render(bundle.desc, camera, p, use_bvh=1)
    b, _ = orc.render(bundle.desc, camera, p, use_bvh=0)
    differing = (np.abs(a - b).max(axis=-1) > 0).mean()
    assert differing < 0.03


def test_strip_ownership_partitions_the_frame(orc):
    bundle, cam, _ = S.cornell_box()
    w, h, spp = 64, 45, 4
    camera = S.camera_for(cam, w, h)
    full, segs = orc.render(bundle.desc, camera, S.abi.render_params(w, h, spp))
    for count, rows in ((2, 8), (3, 4), (8, 8)):
        acc = np.full_like(full, -1.0)
        total = 0
        for idx in range(count):
            part, s = orc.render(bundle.desc, camera, S.abi.render_params(w, h, spp, strip_rows=rows, strip_count=count, strip_index=idx))
            own = ((np.arange(h) // rows) % count) == idx
            assert (part[~own] == 0).all()
            acc[own] = part[own]
            total += s
        assert np.array_equal(acc, full) and total == segs


def test_segments_per_sample_statistics(orc):
    """cornell at 16:9 has ~49 % one-segment paths and ~3.5
    segments per sample; three_balls ~1.9."""
    bundle, cam, _ = S.cornell_box()
    camera = S.camera_for(cam, 160, 90)
    _, segs = orc.render(bundle.desc, camera, S.abi.render_params(160, 90, 8))
    assert 3.3 < segs / (160 * 90 * 8) < 3.7
    bundle, cam, _ = S.three_balls()
    camera = S.camera_for(cam, 160, 90)
    _, segs = orc.render(bundle.desc, camera, S.abi.render_params(160, 90, 8))
    assert 1.7 < segs / (160 * 90 * 8) < 2.1


@pytest.mark.parametrize("name", sorted(fx.SCENES))
def test_oracle_reproduces_committed_golden_frames(orc, host, name):
    """tests/golden/oracle_frames.npz was rendered by this oracle at seed 1; a
    change in the RNG contract, the loader or the oracle shows up here."""
    s, p, use_bvh = fx.load(host, name)
    cam = fx.camera_for(host, s, p)
    frame, segs = orc.render(s.desc, cam, p, use_bvh=use_bvh)
    gold = GOLD[name + "_frame"]
    assert frame.shape == gold.shape
    assert np.allclose(frame, gold, rtol=0, atol=1e-12)
    assert segs == int(GOLD[name + "_segments"])


def test_invalid_arguments(orc):
    bundle, cam, _ = S.two_balls()
    camera = S.camera_for(cam, 8, 8)
    with pytest.raises(RuntimeError):
        orc.render(bundle.desc, camera, S.abi.render_params(8, 8, 0))
    empty = S.abi.SceneBundle([], [], [], S.abi.sky())
    frame, segs = orc.render(empty.desc, camera, S.abi.render_params(8, 8, 2))
    assert segs == 8 * 8 * 2 and (frame > 0.7).all()     # empty scene: background only (SURVEY B-18)


def test_preview_renderer_replicates_block_origins(orc):
    """cpu_scaled.rs:45-98: with scale > 1 only the top-left pixel of each block is
    traced (same draws as that pixel in a full render) and the block is filled with it."""
    bundle, cam, _ = S.three_balls()
    w, h, spp = 100, 45, 3
    camera = S.camera_for(cam, w, h)
    full, _ = orc.render(bundle.desc, camera, S.abi.render_params(w, h, spp, tiles_w=10, tiles_h=5))
    prev, segs = orc.render(bundle.desc, camera, S.abi.render_params(w, h, spp, tiles_w=10, tiles_h=5, scale=4))
    # tile 10 x 9: scale_w = largest divisor of 10 that is <= 4 -> 2; scale_h = largest divisor of 9 <= 4 -> 3
    sw, sh = 2, 3
    assert np.array_equal(prev[::sh, ::sw], full[::sh, ::sw])
    for dy in range(sh):
        for dx in range(sw):
            assert np.array_equal(prev[dy::sh, dx::sw], prev[::sh, ::sw])
    assert segs < 0.25 * w * h * spp * 3
    # a tile grid that does not divide the image: the last column/row tile absorbs the remainder
    # (cpu.rs:97-109) but only whole blocks are drawn, the rest of the tile stays black (cpu_scaled.rs:50-52)
    w2, h2 = 103, 47
    cam2 = S.camera_for(cam, w2, h2)
    prev2, _ = orc.render(bundle.desc, cam2, S.abi.render_params(w2, h2, spp, tiles_w=10, tiles_h=5, scale=4))
    assert (prev2[:, 102:] == 0).all() and (prev2[45:, :] == 0).all()     # 103 = 51*2 + 1, 47 = 15*3 + 2
    assert (prev2[:45, :102] > 0).any()
