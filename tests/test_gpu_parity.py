"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.

Tolerance: BASELINE.json's north star states per-channel |delta| < 1e-3 at a
fixed seed, on the post-tone-map float image.  Both sides compute in f64 with
the same addressed random draws, so the expected difference is ~1e-13; the
tests assert the stated 1e-3 on every pixel AND a much tighter bound on the
bulk, so that a real divergence cannot hide under the tolerance.
"""
import numpy as np
import pytest

import scenes_py as S

pytestmark = pytest.mark.gpu

TOL = 1e-3          # the north star's per-channel tolerance
TIGHT = 1e-9        # what f64-vs-f64 with identical draws should really achieve


def _both(rt, orc, scene_fn, width, height, samples, seed=1, max_depth=20, use_bvh=1, **kw):
    bundle, cam, tm = scene_fn()
    camera = S.camera_for(cam, width, height)
    params = S.abi.render_params(width, height, samples, max_depth=max_depth, seed=seed, **kw)
    ref, ref_segs = orc.render(bundle.desc, camera, params, use_bvh=use_bvh)
    scene = rt.Scene(bundle)
    try:
        got = scene.render_frame(camera, params)
        stats = scene.last_stats()
    finally:
        scene.close()
    kind = {"None": orc.ORC_TM_NONE, "Aces": orc.ORC_TM_ACES}[tm]
    return orc.tone_map(kind, ref), orc.tone_map(kind, got), ref_segs, stats


def _assert_parity(ref, got):
    assert np.isfinite(got).all()
    diff = np.abs(ref - got)
    assert diff.max() < TOL, "max |delta| = %g" % diff.max()
    # bulk must agree to rounding: at most a handful of pixels may carry a
    # branch flip (ulp-level libm / FMA differences at a comparison)
    frac_loose = float((diff.max(axis=-1) > TIGHT).mean())
    assert frac_loose < 1e-3, "fraction of pixels beyond %g: %g" % (TIGHT, frac_loose)


# use_bvh = 0 for the two-box scene: the oracle's BVH reproduces RotateY's
# mis-sized bounding box (rotate_y.rs:66-90, SURVEY B-14) and so culls a thin
# sliver of each box; the device path intersects the true geometry.  The linear
# scan is the reference's own closest-hit semantics without that culling.
@pytest.mark.parametrize("scene_fn,w,h,spp,use_bvh", [
    (S.cornell_box, 160, 90, 32, 1),
    (S.three_balls, 160, 90, 32, 1),
    (S.two_balls, 96, 54, 16, 1),
    (S.cornell_box_boxes, 160, 90, 16, 0),
])
def test_frame_matches_oracle(rt, orc, gpu, scene_fn, w, h, spp, use_bvh):
    ref, got, ref_segs, stats = _both(rt, orc, scene_fn, w, h, spp, use_bvh=use_bvh)
    _assert_parity(ref, got)
    assert stats.samples == w * h * spp
    # same paths => same number of scene.hit evaluations (allow a few flips)
    assert abs(int(stats.segments) - ref_segs) <= max(4, ref_segs // 100000)


def test_baseline_config_1_shape(rt, orc, gpu):
    """BASELINE config 1: three_balls 400x225x16 (non-multiple-of-16 height)."""
    ref, got, ref_segs, stats = _both(rt, orc, S.three_balls, 400, 225, 16)
    _assert_parity(ref, got)


@pytest.mark.parametrize("seed", [2, 3, 0xDEADBEEFCAFE])
def test_seeds(rt, orc, gpu, seed):
    ref, got, _, _ = _both(rt, orc, S.cornell_box, 64, 36, 8, seed=seed)
    _assert_parity(ref, got)


@pytest.mark.parametrize("max_depth", [0, 1, 2, 5])
def test_depth_exhaustion_is_white(rt, orc, gpu, max_depth):
    """renderer.rs:48-55: depth 0 returns (1,1,1), also with tiny max_depth."""
    ref, got, _, _ = _both(rt, orc, S.cornell_box, 64, 36, 8, max_depth=max_depth)
    _assert_parity(ref, got)
    if max_depth == 0:
        assert np.allclose(got, orc.tone_map(orc.ORC_TM_ACES, np.ones_like(got)))


def test_odd_sizes(rt, orc, gpu):
    for w, h in ((17, 9), (33, 31), (2, 2)):
        ref, got, _, _ = _both(rt, orc, S.three_balls, w, h, 4)
        _assert_parity(ref, got)


def test_strips_assemble_to_full_frame(rt, orc, gpu):
    """Multi-GPU row ownership: N strip renders == one full render, bit for bit
    (the RNG is keyed by the global pixel index)."""
    bundle, cam, _ = S.cornell_box()
    w, h, spp = 96, 70, 72   # 72 spp = three sample chunks: the per-chunk sums must not depend on the strip layout
    camera = S.camera_for(cam, w, h)
    scene = rt.Scene(bundle)
    try:
        full = scene.render_frame(camera, S.abi.render_params(w, h, spp))
        again = scene.render_frame(camera, S.abi.render_params(w, h, spp))
        assert np.array_equal(full, again)   # LDS accumulation order is fixed within an item
        for count, rows in ((2, 8), (3, 8), (8, 8), (4, 16)):
            acc = np.full_like(full, -1.0)
            for idx in range(count):
                p = S.abi.render_params(w, h, spp, strip_rows=rows, strip_count=count, strip_index=idx)
                part = scene.render_frame(camera, p)
                own = ((np.arange(h) // rows) % count) == idx
                assert (part[~own] == 0).all()  # unowned rows untouched in the zeroed buffer
                acc[own] = part[own]
            assert np.array_equal(acc, full)
    finally:
        scene.close()


def test_tile_stream_matches_frame(rt, orc, gpu):
    """rt_render emits cpu.rs:73-115's tile grid; stitched tiles == the frame."""
    bundle, cam, _ = S.three_balls()
    w, h, spp = 100, 45, 4
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp, tiles_w=10, tiles_h=10)
    scene = rt.Scene(bundle)
    try:
        frame = scene.render_frame(camera, params)
        tiles = scene.render_tiles(camera, params)
    finally:
        scene.close()
    assert len(tiles) == 100
    # column-major order, last row absorbs the remainder (45 = 9*4 + 9)
    assert [(t[0], t[1]) for t in tiles[:3]] == [(0, 0), (4, 0), (8, 0)]
    assert tiles[9][3] == 9 and tiles[9][2] == 10
    stitched = np.zeros_like(frame)
    for r, c, tw, th, arr in tiles:
        stitched[r:r + th, c:c + tw] = arr
    assert np.array_equal(stitched, frame)


def test_cancel_before_start_returns_cancel_event(rt, orc, gpu):
    import ctypes as C
    bundle, cam, _ = S.two_balls()
    camera = S.camera_for(cam, 32, 18)
    params = S.abi.render_params(32, 18, 2, tiles_w=2, tiles_h=2)
    scene = rt.Scene(bundle)
    try:
        flag = C.c_int(1)
        with pytest.raises(rt.RtError) as e:
            scene.render_tiles(camera, params, cancel=C.pointer(flag))
        assert e.value.code == S.abi.RT_ERR_CANCEL_EVENT  # cpu.rs:82-85
    finally:
        scene.close()


def test_cancel_during_render_returns_ok_and_stops_the_tile_stream(rt, orc, gpu):
    """cpu.rs:55-62: a cancel seen while rendering makes render() return Ok(());
    tiles that finished before it stay written, nothing is written afterwards.
    rt_render delivers tile columns as they finish; the host polls the flag
    before every callback and while it waits, the waves in flight stop at their
    next item once it is up."""
    import ctypes as C
    import threading
    import time
    bundle, cam, _ = S.cornell_box()
    w, h, spp = 1920, 1080, 8192            # ~0.75 s of GPU work in 10 tile columns
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)
    scene = rt.Scene(bundle)
    try:
        scene.render_frame(camera, S.abi.render_params(w, h, 1))   # warm-up (allocations)
        flag = C.c_int(0)
        total = w * h * spp
        # The flag rises at a DETERMINISTIC point — inside the callback of the first delivered tile (the first tile
        # column has just been published) — and what is asserted is counted on the device, not read off a clock:
        # nothing is delivered after the tile whose callback raised the flag (the hook is polled before every
        # callback), and the launch stopped early: of the frame's primary rays only the first column (10 %), the items
        # the 6 144 resident waves had in hand (1.2 %) and what they fetched until the poison landed were started.
        # The time is printed, not asserted (a shared box).
        tiles, raised = [], []

        def on_tile(_user, rgb, r, c, tw, th):
            tiles.append((r, c, tw, th))
            if not raised:
                raised.append(time.time())
                flag.value = 1

        cb = S.abi.RtTileCallback(on_tile)
        rt.check(rt.lib().rt_render(scene._h, C.byref(camera), C.byref(params), cb, None, C.pointer(flag)), "rt_render")
        returned = time.time()
        started = int(scene.last_stats().samples)
        print("rt_render returned %.1f ms after the flag rose; %.1f %% of the frame's primary rays were started"
              % ((returned - raised[0]) * 1e3, 100.0 * started / total))
        assert len(tiles) == 1                                      # the stream stopped with the tile that raised the flag
        assert 0.10 * total <= started < 0.25 * total               # ... and the GPU within an item's time of it
        # what was delivered is a prefix of the column-major tile list (cpu.rs:91-113)
        expect = [(108 * hs, 192 * ws) for ws in range(10) for hs in range(10)]
        assert [(t[0], t[1]) for t in tiles] == expect[:len(tiles)]
        # the scene is still usable afterwards (the poisoned item counters are reset by the next call)
        flag.value = 0
        small = S.abi.render_params(64, 36, 4, tiles_w=2, tiles_h=2)
        small_cam = S.camera_for(cam, 64, 36)
        after = scene.render_tiles(small_cam, small, cancel=C.pointer(flag))
        assert len(after) == 4
        fresh = rt.Scene(bundle)
        try:
            whole = fresh.render_frame(small_cam, small)
        finally:
            fresh.close()
        for r, c, tw, th, arr in after:
            assert np.array_equal(arr, whole[r:r + th, c:c + tw])
        assert np.array_equal(scene.render_frame(small_cam, small), whole)
        # the whole-frame path (a single tile column is cut from the finished frame) stops the same way
        # (here the flag can only rise from another thread: the callbacks come after the frame.  Asserted: no tile, and
        # the device-counted rays say the launch was cut short; the time is printed)
        flag.value = 0
        del raised[:]

        def raise_flag():
            raised.append(time.time())
            flag.value = 1

        timer = threading.Timer(0.05, raise_flag)
        timer.start()
        one_column = S.abi.render_params(w, h, spp, tiles_w=1, tiles_h=4)
        tiles = scene.render_tiles(camera, one_column, cancel=C.pointer(flag))
        returned = time.time()
        timer.join()
        started = int(scene.last_stats().samples)
        print("one column: returned %.1f ms after the flag rose; %.1f %% of the primary rays were started"
              % ((returned - raised[0]) * 1e3, 100.0 * started / total))
        assert tiles == [] and started < total                      # cpu.rs:55-62: Ok(()), no tile written, the launch cut short
        flag.value = 0
        assert np.array_equal(scene.render_frame(small_cam, small), whole)
    finally:
        scene.close()


@pytest.mark.parametrize("w,h,tw,th", [(101, 47, 7, 3), (64, 36, 2, 2), (50, 30, 1, 4), (37, 23, 10, 10), (9, 9, 12, 2), (20, 5, 2, 8)])
def test_progressive_tiles_are_bit_identical_to_the_frame(rt, orc, gpu, w, h, tw, th):
    """Tile columns are regions of ONE delivering launch (column windows of the item grid,
    racer-tracer_amd/csrc/rt_deliver.hip) and are handed over while the next column renders;
    every pixel must still be the whole-frame render's pixel bit for bit, in cpu.rs:73-115's
    order, with and without a cancel flag.  Empty tiles (a grid with more rows or columns than the image has pixels)
    are sent too, as empty BufferUpdates: the reference's raytrace sends every SubImage's buffer (cpu.rs:64-70)."""
    import ctypes as C
    bundle, cam, _ = S.cornell_box_boxes()
    spp = 40
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp, tiles_w=tw, tiles_h=th)
    scene = rt.Scene(bundle)
    try:
        frame = scene.render_frame(camera, params)
        seg_frame = scene.last_stats().segments
        for flag in (None, C.c_int(0)):
            tiles = scene.render_tiles(camera, params, cancel=None if flag is None else C.pointer(flag))
            st = scene.last_stats()
            ws_step, hs_step = w // tw, h // th
            expect = []
            for ws in range(tw):
                for hs in range(th):
                    tile_w = w - ws_step * ws if ws == tw - 1 else ws_step
                    tile_h = h - hs_step * hs if hs == th - 1 else hs_step
                    expect.append((hs_step * hs, ws_step * ws, max(tile_w, 0), max(tile_h, 0)))
            assert [t[:4] for t in tiles] == expect and len(tiles) == tw * th
            stitched = np.full_like(frame, -1.0)
            for r, c, tile_w, tile_h, arr in tiles:
                stitched[r:r + tile_h, c:c + tile_w] = arr
            assert np.array_equal(stitched, frame)
            assert st.samples == w * h * spp and st.segments == seg_frame
    finally:
        scene.close()


def test_light_linearity_at_full_size(rt, orc, gpu):
    """Size-independent property at BASELINE's frame size: every cornell_box path ends in the
    light or in black, so doubling the light's emission doubles each pixel's radiance sum
    exactly (a power-of-two scale) — the gamma-encoded frames differ by sqrt(2)."""
    bundle, cam, _ = S.cornell_box()
    brighter, _, _ = S.cornell_box()
    brighter.textures[3].color = S.abi.D3(30.0, 30.0, 30.0)
    w, h, spp = 1920, 1080, 8
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)                   # (depth exhaustion adds white, which does not scale: ~1 % of paths)
    frames = []
    for b in (bundle, brighter):
        scene = rt.Scene(b)
        try:
            frames.append(scene.render_frame(camera, params))
        finally:
            scene.close()
    a2, b2 = frames[0] ** 2, frames[1] ** 2
    lit = a2 > 0
    assert lit.mean() > 0.1                                    # at 8 spp most in-box pixels are still black
    white_only = np.isclose(a2, b2) & lit                      # paths that ended by depth exhaustion only
    scaled = np.isclose(b2, 2.0 * a2, rtol=1e-12, atol=0)
    mixed = lit & ~scaled & ~white_only                        # pixels mixing both kinds of path ends
    assert scaled.sum() > 0.5 * lit.sum()
    assert np.all(b2[mixed] > a2[mixed]) and np.all(b2[mixed] < 2.0 * a2[mixed] * (1 + 1e-12))


def test_full_size_properties(rt, orc, gpu):
    """BASELINE config 3 size (1920x1080) at reduced spp: properties that do
    not need the oracle at full size + an oracle check on a row band."""
    bundle, cam, _ = S.cornell_box()
    w, h, spp = 1920, 1080, 4
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)
    scene = rt.Scene(bundle)
    try:
        a = scene.render_frame(camera, params)
        b = scene.render_frame(camera, params)
        stats = scene.last_stats()
    finally:
        scene.close()
    assert np.array_equal(a, b)                      # deterministic
    assert np.isfinite(a).all() and (a >= 0).all()
    assert (a[:, :300] == 0).all() and (a[:, -300:] == 0).all()  # black outside the box
    assert stats.samples == w * h * spp
    assert 3.0 < stats.segments / stats.samples < 4.0  # SURVEY: ~3.5 segments/sample
    # oracle on a band of rows through the light (strip ownership keeps it cheap)
    band = S.abi.render_params(w, h, spp, strip_rows=8, strip_count=135, strip_index=20)
    ref, _ = orc.render(bundle.desc, camera, band)
    rows = ((np.arange(h) // 8) % 135) == 20
    d = np.abs(ref[rows] - a[rows])
    assert d.max() < TOL
    assert float((d.max(axis=-1) > TIGHT).mean()) < 1e-3
