"""The delivering launch (racer-tracer_amd/csrc/rt_deliver.hip): rt_render_frame, rt_render, rt_render_ex and the
several-device forms finish their own pixels inside ONE persistent launch and write them straight into pinned host
memory.  Everything here is a bit-for-bit comparison against the two-pass path (trace + resolve kernel into device
memory: rt_render_frame_device, what bench.py times), which the parity tests hold against the oracle.

Reference behaviour being replaced: CpuRenderer::render's tile stream (renderer/cpu.rs:64-70,118-131) and its cancel
hook `do_cancel` (renderer.rs:25-30), a `SignalEvent::wait_timeout(0)` poll — bound here by rt_render_ex's callback."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

import scenes_py as S

pytestmark = pytest.mark.gpu


def two_pass_frame(scene, camera, params):
    """rt_render_frame_device into a torch buffer: trace kernel + resolve kernel, no delivery."""
    import torch
    out = torch.zeros((params.height, params.width, 3), dtype=torch.float64, device="cuda")
    scene.render_frame_device(camera, params, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("scene_name,w,h,spp", [("cornell_box", 200, 120, 40), ("three_balls", 203, 117, 24),
                                                 ("noise_and_textures", 160, 90, 16), ("cornell_box_boxes", 96, 54, 300)])
def test_delivered_frame_is_the_two_pass_frame(rt, gpu, scene_name, w, h, spp):
    if scene_name == "noise_and_textures":  # textures come through the YAML loader (Perlin table, earth map); 16:9 like its config
        import importlib
        import os
        host = importlib.import_module("racer-tracer_amd.host")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        bundle = host.Session(os.path.join(root, "scenes", "config_c4.yml"), scene=os.path.join(root, "scenes", "noise_and_textures.yml"))
        camera = bundle.camera
    else:
        bundle, cam, _ = getattr(S, scene_name)()
        camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)
    scene = rt.Scene(bundle)
    try:
        want = two_pass_frame(scene, camera, params)
        segs = scene.last_stats().segments
        for _ in range(3):  # the counters re-arm themselves: every call must deliver every pixel again
            got = scene.render_frame(camera, params)
            assert np.array_equal(got, want)
            assert scene.last_stats().segments == segs
        # strips: owned rows only, the rest untouched
        for count, rows in ((2, 8), (3, 16), (5, 8)):
            acc = np.full_like(want, -1.0)
            for idx in range(count):
                p = S.abi.render_params(w, h, spp, strip_rows=rows, strip_count=count, strip_index=idx)
                part = scene.render_frame(camera, p)
                own = ((np.arange(h) // rows) % count) == idx
                assert (part[~own] == 0).all()
                acc[own] = part[own]
            assert np.array_equal(acc, want)
    finally:
        scene.close()


def test_delivery_at_full_size_many_chunks(rt, gpu):
    """BASELINE config 3's frame at 256 spp (16 chunks + taper): 32 400 item tiles, each finished by whichever wave
    lands its last chunk — across all eight XCDs, whose L2s are not coherent with each other."""
    bundle, cam, _ = S.cornell_box()
    w, h, spp = 1920, 1080, 256
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)
    scene = rt.Scene(bundle)
    try:
        want = two_pass_frame(scene, camera, params)
        for _ in range(2):
            assert np.array_equal(scene.render_frame(camera, params), want)
        tiles = scene.render_tiles(camera, params)
        assert len(tiles) == 100
        stitched = np.full_like(want, -1.0)
        for r, c, tw, th, arr in tiles:
            stitched[r:r + th, c:c + tw] = arr
        assert np.array_equal(stitched, want)
    finally:
        scene.close()


@pytest.mark.parametrize("w,h,tw,th", [(100, 45, 40, 3), (333, 64, 64, 2), (64, 36, 33, 1)])
def test_more_tile_columns_than_regions(rt, gpu, w, h, tw, th):
    """A launch queues at most 32 regions: wider tile grids share regions, the tiles stay cpu.rs:73-115's."""
    bundle, cam, _ = S.cornell_box_boxes()
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, 12, tiles_w=tw, tiles_h=th)
    scene = rt.Scene(bundle)
    try:
        frame = two_pass_frame(scene, camera, params)
        tiles = scene.render_tiles(camera, params)
    finally:
        scene.close()
    ws_step, hs_step = w // tw, h // th
    expect = [(hs_step * hs, ws_step * ws, w - ws_step * ws if ws == tw - 1 else ws_step, h - hs_step * hs if hs == th - 1 else hs_step)
              for ws in range(tw) for hs in range(th)]
    assert [t[:4] for t in tiles] == expect
    stitched = np.full_like(frame, -1.0)
    for r, c, tile_w, tile_h, arr in tiles:
        stitched[r:r + tile_h, c:c + tile_w] = arr
    assert np.array_equal(stitched, frame)


def test_cancel_callback_binds_like_do_cancel(rt, gpu):
    """rt_render_ex: the cancel hook is a FUNCTION (renderer.rs:25-30 polls SignalEvent::wait_timeout(0)); a
    threading.Event stands in for the SignalEvent.  Never raised: all tiles.  Raised mid-render: RT_OK, a prefix
    of the tile list, the GPU drained within an item's time; raised on entry: RT_ERR_CANCEL_EVENT (cpu.rs:82-85)."""
    bundle, cam, _ = S.cornell_box()
    w, h = 1920, 1080
    camera = S.camera_for(cam, w, h)
    scene = rt.Scene(bundle)
    try:
        event = threading.Event()
        polls = []

        def cancelled():
            polls.append(1)
            return event.is_set()

        quick = S.abi.render_params(w, h, 16)
        tiles = scene.render_tiles(camera, quick, cancel=cancelled)
        assert len(tiles) == 100 and len(polls) >= 100            # polled before every tile at least
        want = two_pass_frame(scene, camera, quick)
        for r, c, tw, th, arr in tiles:
            assert np.array_equal(arr, want[r:r + th, c:c + tw])

        long_render = S.abi.render_params(w, h, 8192)               # ~0.65 s of GPU work
        raised = []

        def raise_it():
            raised.append(time.time())
            event.set()

        timer = threading.Timer(0.05, raise_it)
        timer.start()
        tiles = scene.render_tiles(camera, long_render, cancel=cancelled)
        returned = time.time()
        timer.join()
        assert len(tiles) < 100
        expect = [(108 * hs, 192 * ws) for ws in range(10) for hs in range(10)]
        assert [(t[0], t[1]) for t in tiles] == expect[:len(tiles)]
        started = int(scene.last_stats().samples)
        print("rt_render_ex returned %.1f ms after the event was set; %.1f %% of the primary rays were started"
              % ((returned - raised[0]) * 1e3, 100.0 * started / (w * h * 8192)))
        assert started < w * h * 8192                               # counted on the device: the launch was cut short (the time is printed, not asserted)
        with pytest.raises(rt.RtError) as e:
            scene.render_tiles(camera, quick, cancel=cancelled)     # still set
        assert e.value.code == S.abi.RT_ERR_CANCEL_EVENT
        event.clear()                                               # the scene is reusable: counters were cleared
        assert np.array_equal(scene.render_frame(camera, quick), want)
    finally:
        scene.close()


def test_tile_stream_over_several_scenes_on_one_card(rt, gpu):
    """rt_render_multi with 1, 2 and 3 shares on device 0 (a device may be listed twice): every share traces its
    interleaved 8-row strips, all of them write into ONE pinned frame, and a tile column is handed over once every
    share has published it.  Same tiles, same order, same pixels as rt_render; cancel works the same."""
    bundle, cam, _ = S.three_balls()
    w, h, spp = 330, 200, 24
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp, tiles_w=7, tiles_h=3)
    scenes = [rt.Scene(bundle) for _ in range(3)]
    try:
        want_tiles = scenes[0].render_tiles(camera, params)
        for n in (1, 2, 3):
            tiles = rt.render_tiles_multi(scenes[:n], camera, params)
            assert [t[:4] for t in tiles] == [t[:4] for t in want_tiles]
            for got, want in zip(tiles, want_tiles):
                assert np.array_equal(got[4], want[4])
            total = sum(int(s.last_stats().samples) for s in scenes[:n])
            assert total == w * h * spp
        big = S.abi.render_params(1920, 1080, 4096, tiles_w=10, tiles_h=10)
        big_cam = S.camera_for(cam, 1920, 1080)
        event = threading.Event()
        timer = threading.Timer(0.05, event.set)
        timer.start()
        t0 = time.time()
        tiles = rt.render_tiles_multi(scenes[:2], big_cam, big, cancel=event.is_set)
        timer.join()
        print("rt_render_multi (2 shares) returned %.1f ms after the call began" % ((time.time() - t0) * 1e3))
        started = sum(int(sc.last_stats().samples) for sc in scenes[:2])
        # counted on the device: the share that was running stopped within an item's time of the hook (it had started
        # under half of its rays), the one still waiting behind it on this one card started none
        assert len(tiles) < 100 and started < 0.5 * 1920 * 1080 * 4096
        after = rt.render_tiles_multi(scenes[:2], camera, params)
        for got, want in zip(after, want_tiles):
            assert np.array_equal(got[4], want[4])
        with pytest.raises(rt.RtError):
            rt.render_tiles_multi([scenes[0], scenes[0]], camera, params)   # the same scene twice
    finally:
        for s in scenes:
            s.close()


def test_cancel_reaches_a_launch_that_fills_the_register_file(rt, gpu):
    """The textured variants run four waves of 128 VGPRs per SIMD: nothing else fits on the GPU while they run.  The
    cancel must not depend on anything finding room there (rounds 2-3 wrote the item counters with
    hipStreamWriteValue32, a small kernel of the runtime's: it landed when the launch was over); the waves read a word in
    pinned host memory instead (TraceArgs.cancel_flag).  noise_and_textures at 4096 spp is ~0.42 s of GPU work; the hook
    rises from the first tile's callback and the launch must have started well under half of its rays (counted on the
    device) — like do_cancel stopping the reference's tile loop (renderer.rs:25-30, cpu.rs:55-62)."""
    import importlib
    import os
    host = importlib.import_module("racer-tracer_amd.host")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    session = host.Session(os.path.join(root, "scenes", "config_c4.yml"), scene=os.path.join(root, "scenes", "noise_and_textures.yml"))
    p = session.params
    p.samples = 4096
    total = p.width * p.height * p.samples
    scene = rt.Scene(session)
    try:
        scene.render_frame(session.camera, S.abi.render_params(p.width, p.height, 1))   # allocations out of the way
        flag = C.c_int(0)
        tiles = []

        def on_tile(_user, rgb, r, c, tw, th):
            tiles.append((r, c))
            flag.value = 1

        cb = S.abi.RtTileCallback(on_tile)
        t0 = time.time()
        rt.check(rt.lib().rt_render(scene._h, C.byref(session.camera), C.byref(p), cb, None, C.pointer(flag)), "rt_render")
        started = int(scene.last_stats().samples)
        print("noise_and_textures: rt_render returned after %.1f ms with %.1f %% of the primary rays started" % ((time.time() - t0) * 1e3, 100.0 * started / total))
        assert len(tiles) == 1 and 0.10 * total <= started < 0.30 * total
    finally:
        scene.close()


def test_a_cancel_hook_that_never_fires_costs_nothing(rt, gpu):
    """The reference's callers always pass a cancel event (main.rs:163-199); the hook must not tax the render.  The
    waves' look at the cancel word is a read of pinned HOST memory: asked with every work item it cost the C3 tile
    stream 30 % (the link serves ~28 M such reads a second), so one item in 32 asks.  Compared on the device's own
    clock (HIP events around the launch): the same frame with and without a hook, frames equal."""
    bundle, cam, _ = S.cornell_box()
    w, h, spp = 1920, 1080, 256
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)
    scene = rt.Scene(bundle)
    try:
        scene.render_tiles(camera, params)                        # warm-up
        ms = {}
        frames = {}
        for name, hook in (("plain", None), ("hook", lambda: False), ("plain2", None), ("hook2", lambda: False)):
            tiles = scene.render_tiles(camera, params, cancel=hook)
            ms[name] = scene.last_stats().kernel_ms
            frames[name] = np.concatenate([t[4].ravel() for t in tiles])
        print("kernel ms without / with a cancel hook: %.2f / %.2f, %.2f / %.2f" % (ms["plain"], ms["hook"], ms["plain2"], ms["hook2"]))
        assert np.array_equal(frames["plain"], frames["hook"])
        assert min(ms["hook"], ms["hook2"]) < 1.15 * min(ms["plain"], ms["plain2"])   # (asked with every item: 1.30; run-to-run spread on a shared box: 0.04)
    finally:
        scene.close()


def test_whole_frame_over_several_scenes_matches(rt, gpu):
    bundle, cam, _ = S.cornell_box()
    w, h, spp = 320, 180, 48
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)
    scenes = [rt.Scene(bundle) for _ in range(3)]
    try:
        want = two_pass_frame(scenes[0], camera, params)
        for n in (1, 2, 3):
            for rows in (0, 16):
                assert np.array_equal(rt.render_frame_multi(scenes[:n], camera, params, strip_rows=rows), want)
    finally:
        for s in scenes:
            s.close()


def test_shares_that_own_no_rows(rt, gpu):
    """More shares than 8-row strips (height 16 over three scenes), or a strip_index whose first strip lies below the
    image: such a share launches nothing and counts as published; the call succeeds, the other rows are complete and
    the share's (non-existent) rows of the caller's buffer stay untouched — what the two-pass path does with zero work
    items."""
    bundle, cam, _ = S.cornell_box()
    w, h, spp = 64, 16, 12
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp, tiles_w=4, tiles_h=2)
    scenes = [rt.Scene(bundle) for _ in range(4)]
    try:
        want = two_pass_frame(scenes[0], camera, params)
        want_tiles = scenes[0].render_tiles(camera, params)
        for n in (3, 4):                                             # 2 strips: shares 2 and 3 own nothing
            assert np.array_equal(rt.render_frame_multi(scenes[:n], camera, params), want)
            tiles = rt.render_tiles_multi(scenes[:n], camera, params)
            assert [t[:4] for t in tiles] == [t[:4] for t in want_tiles]
            for got, ref in zip(tiles, want_tiles):
                assert np.array_equal(got[4], ref[4])
            assert [int(s.last_stats().samples) for s in scenes[:2]] == [w * 8 * spp] * 2
        # one scene, a strip index beyond the image: RT_OK, nothing written (render_frame hands in a zeroed buffer)
        for idx in (2, 4):
            p = S.abi.render_params(w, h, spp, strip_rows=8, strip_count=5, strip_index=idx)
            assert (scenes[0].render_frame(camera, p) == 0).all()
        p = S.abi.render_params(w, h, spp, strip_rows=8, strip_count=5, strip_index=1)
        part = scenes[0].render_frame(camera, p)                     # ... and the scene still renders afterwards
        assert np.array_equal(part[8:], want[8:]) and (part[:8] == 0).all()
    finally:
        for s in scenes:
            s.close()


def test_render_buffers_survive_scene_rebuilds(rt, gpu):
    """The reference rebuilds its scene on every object event (main.rs:174-189); rt_scene_destroy hands what a render
    allocated (slices, pinned frame, flags, counters, streams) to the next rt_scene_create on the device.  A rebuilt
    scene — another description, another frame size, delivering and two-pass calls mixed — must render exactly what a
    scene with fresh buffers renders: the host-visible flags still hold the serials of the previous owner, the tile
    counters must be at zero, the slices may be larger or smaller than the new frame needs."""
    cases = [(S.cornell_box, 200, 120, 40), (S.three_balls, 96, 54, 8), (S.cornell_box_boxes, 333, 64, 12), (S.cornell_box, 200, 120, 40)]
    rt.lib().rt_release_cached_buffers()                     # whatever earlier tests left behind
    fresh = []
    for make, w, h, spp in cases:                             # every scene on buffers of its own
        bundle, cam, _ = make()
        scene = rt.Scene(bundle)
        fresh.append(two_pass_frame(scene, S.camera_for(cam, w, h), S.abi.render_params(w, h, spp)))
        scene.close()
        rt.lib().rt_release_cached_buffers()
    for round_ in range(2):
        for (make, w, h, spp), want in zip(cases, fresh):     # ... and now handing them down from scene to scene
            bundle, cam, _ = make()
            camera, params = S.camera_for(cam, w, h), S.abi.render_params(w, h, spp, tiles_w=5, tiles_h=3)
            scene = rt.Scene(bundle)
            try:
                assert np.array_equal(scene.render_frame(camera, params), want)
                stitched = np.full_like(want, -1.0)
                for r, c, tw, th, arr in scene.render_tiles(camera, params):
                    stitched[r:r + th, c:c + tw] = arr
                assert np.array_equal(stitched, want)
                assert np.array_equal(two_pass_frame(scene, camera, params), want)
            finally:
                scene.close()
    a, b = rt.Scene(cases[0][0]()[0]), rt.Scene(cases[1][0]()[0])   # two alive at once: two sets, no sharing
    try:
        for sc, (make, w, h, spp), want in ((a, cases[0], fresh[0]), (b, cases[1], fresh[1])):
            assert np.array_equal(sc.render_frame(S.camera_for(make()[1], w, h), S.abi.render_params(w, h, spp)), want)
    finally:
        a.close()
        b.close()
    rt.lib().rt_release_cached_buffers()


def test_paths_that_do_not_deliver_still_stream_tiles(rt, gpu):
    """The v1 kernel and the preview scale render with the two-pass path and cut the tiles from the finished frame
    (rt_deliver.hip: tiles_from_frame): same tiles, the cancel callback honoured; several devices refuse them."""
    bundle, cam, _ = S.three_balls()
    w, h = 96, 54
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, 6, tiles_w=4, tiles_h=3)
    v1 = rt.Scene(bundle, kernel=S.abi.RT_KERNEL_V1)
    pool = rt.Scene(bundle)
    try:
        frame = v1.render_frame(camera, params)
        polls = []
        tiles = v1.render_tiles(camera, params, cancel=lambda: polls.append(1) and False)
        assert len(tiles) == 12 and len(polls) >= 12
        for r, c, tw, th, arr in tiles:
            assert np.array_equal(arr, frame[r:r + th, c:c + tw])
        count = [0]

        def fires_on_the_sixth_poll():   # not on entry (that would be RT_ERR_CANCEL_EVENT): somewhere inside the call
            count[0] += 1
            return count[0] > 5

        cut = v1.render_tiles(camera, params, cancel=fires_on_the_sixth_poll)
        assert len(cut) < 12 and [t[:4] for t in cut] == [t[:4] for t in tiles[:len(cut)]]
        with pytest.raises(rt.RtError) as e:
            rt.render_tiles_multi([v1, pool], camera, params)
        assert e.value.code == S.abi.RT_ERR_UNSUPPORTED
        preview = S.abi.render_params(w, h, 6, tiles_w=4, tiles_h=3, scale=3)
        with pytest.raises(rt.RtError):
            rt.render_tiles_multi([pool, rt.Scene(bundle)], camera, preview)
    finally:
        v1.close()
        pool.close()


@pytest.mark.parametrize("rows,count", [(5, 3), (12, 2), (3, 4)])
def test_delivery_with_strips_that_cut_item_tiles(rt, gpu, rows, count):
    """Strip heights that are not a multiple of 8 regroup the pixels that share an item tile (rt_abi.h): the frame then
    agrees with the whole-frame render to rounding only — but the delivering launch and the two-pass path still see the
    same tiles, so THEY must agree bit for bit, share by share (deliver_item's per-lane row mapping)."""
    import torch
    bundle, cam, _ = S.cornell_box_boxes()
    w, h, spp = 150, 83, 20
    camera = S.camera_for(cam, w, h)
    scene = rt.Scene(bundle)
    try:
        whole = two_pass_frame(scene, camera, S.abi.render_params(w, h, spp))
        for idx in range(count):
            p = S.abi.render_params(w, h, spp, strip_rows=rows, strip_count=count, strip_index=idx)
            dev = torch.full((h, w, 3), -1.0, dtype=torch.float64, device="cuda")
            scene.render_frame_device(camera, p, dev.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            want = dev.cpu().numpy()
            got = scene.render_frame(camera, p)
            own = ((np.arange(h) // rows) % count) == idx
            assert np.array_equal(got[own], want[own])
            assert (got[~own] == 0).all() and (want[~own] == -1).all()
            assert np.abs(got[own] - whole[own]).max() < 1e-12
    finally:
        scene.close()
