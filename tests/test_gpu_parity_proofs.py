"""Parity proofs that go beyond "GPU == oracle on a small frame":

* closest hit at 20 000 spheres without chaos (max_depth 1, one colour per sphere): any defect of the
  global-memory BVH walk shows as a wrong colour, with the full per-pixel tolerance;
* the same chaotic hall of mirrors as test_gpu_edges, through RtSceneOptions.arithmetic = RT_ARITH_REFERENCE
  (IEEE divisions, no FMA contraction — the reference's own operations): with the arithmetic difference
  removed the full tolerance must hold, which proves the product build's outliers are arithmetic
  (1-2 ulp of the reciprocal-based divisions amplified by D/r per bounce), not traversal;
* BASELINE configs 2 and 4 at FULL size and FULL spp, config 5's per-GPU share at 4096 spp (sample chunks
  of 64), each against an oracle band;
* several devices in one process (rt_render_frame_multi*), statistics counted on the device.
"""
import importlib
import os

import numpy as np
import pytest

import scenes_py as S

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-3      # north star: per-channel |delta| < 1e-3
TIGHT = 1e-9


@pytest.fixture(scope="module")
def host():
    return importlib.import_module("racer-tracer_amd.host")


@pytest.fixture(scope="module")
def exact():
    """RtSceneOptions.arithmetic = RT_ARITH_REFERENCE: what any caller of the C ABI can ask for."""
    return S.abi.RT_ARITH_REFERENCE


def hall_of_spheres(n, one_colour_each):
    abi = S.abi
    rng = np.random.default_rng(5)
    centers = rng.uniform(-40.0, 40.0, size=(n, 3))
    centers[:, 2] = rng.uniform(-90.0, -10.0, size=n)
    radii = rng.uniform(0.2, 0.9, size=n)
    if one_colour_each:
        colours = rng.uniform(0.05, 0.95, size=(n, 3))
        textures = [abi.solid(tuple(c)) for c in colours]
        materials = [abi.material(abi.RT_MAT_LAMBERTIAN, i) for i in range(n)]
        prims = [abi.sphere(tuple(centers[i]), float(radii[i]), i, i) for i in range(n)]
    else:
        textures = [abi.solid((0.8, 0.3, 0.3)), abi.solid((0.3, 0.8, 0.3)), abi.solid((0.9, 0.9, 0.9))]
        materials = [abi.material(abi.RT_MAT_LAMBERTIAN, 0), abi.material(abi.RT_MAT_LAMBERTIAN, 1),
                     abi.material(abi.RT_MAT_METAL, 2, fuzz=0.1), abi.material(abi.RT_MAT_DIELECTRIC, -1, ior=1.5)]
        prims = [abi.sphere(tuple(centers[i]), float(radii[i]), int(i % 4), i) for i in range(n)]
    bundle = abi.SceneBundle(prims, materials, textures, abi.sky())
    cam = dict(look_from=(0.0, 0.0, 5.0), look_at=(0.0, 0.0, -50.0), vfov=50.0, aperture=0.0, focus_distance=10.0)
    return bundle, cam


def test_closest_hit_at_20000_spheres_without_chaos(rt, orc, gpu):
    """max_depth 1: ray_color(depth 1) = albedo of the FIRST hit x white (renderer.rs:48-55), or the sky.
    Every sphere has its own colour, so a wrong-but-existing hit (a skipped subtree, a wrong leaf remap,
    a t_near/t_far slip) cannot hide: it is a different colour.  Checked against the oracle's linear scan
    and against the oracle's reference-shaped BVH, with the full tolerance on every pixel."""
    n = 20000
    bundle, cam = hall_of_spheres(n, one_colour_each=True)
    w, h, spp = 160, 90, 4
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp, max_depth=1)
    scene = rt.Scene(bundle)
    try:
        got = scene.render_frame(camera, params)
        stats = scene.last_stats()
    finally:
        scene.close()
    assert stats.segments == w * h * spp == stats.samples      # one level per sample, both counted on the device
    for use_bvh in (0, 1):
        ref, _ = orc.render(bundle.desc, camera, params, use_bvh=use_bvh)
        d = np.abs(got - ref)
        assert d.max() < TOL, (use_bvh, d.max())
        assert (d > TIGHT).mean() < 1e-3, use_bvh
    sky = np.all(np.abs(got - np.sqrt([0.75, 0.85, 1.0])) < 0.13, axis=-1)
    assert 0.05 < 1.0 - sky.mean() < 0.999 and len(np.unique(np.round(got.reshape(-1, 3), 6), axis=0)) > 2000


@pytest.mark.parametrize("n", [20000])
def test_hall_of_mirrors_in_exact_arithmetic(rt, orc, gpu, exact, n):
    """The scene whose per-pixel tolerance test_gpu_edges has to loosen for the product build, through the
    build that divides like the reference does: full tolerance, every pixel."""
    bundle, cam = hall_of_spheres(n, one_colour_each=False)
    w, h, spp = 96, 54, 3
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp, max_depth=8)
    ref, ref_segs = orc.render(bundle.desc, camera, params)
    scene = rt.Scene(bundle, arithmetic=exact)
    try:
        got = scene.render_frame(camera, params)
        stats = scene.last_stats()
    finally:
        scene.close()
    d = np.abs(got - ref)
    assert np.isfinite(got).all() and got.std() > 0.05
    assert int(stats.segments) == ref_segs
    assert d.max() < TOL, d.max()
    assert (d > TIGHT).mean() < 1e-3


@pytest.mark.parametrize("scene_fn", [S.three_balls, S.cornell_box_boxes])
def test_reference_arithmetic_agrees_with_fast_arithmetic(rt, orc, gpu, exact, scene_fn):
    """The two copies of the kernels differ only in rounding: same paths, frames equal to ~1e-12 on the shipped scenes."""
    bundle, cam, _ = scene_fn()
    camera = S.camera_for(cam, 96, 54)
    params = S.abi.render_params(96, 54, 8)
    frames = []
    for arithmetic in (S.abi.RT_ARITH_FAST, exact):
        scene = rt.Scene(bundle, arithmetic=arithmetic)
        try:
            frames.append(scene.render_frame(camera, params))
        finally:
            scene.close()
    ref, _ = orc.render(bundle.desc, camera, params, use_bvh=0)
    if scene_fn is S.three_balls:
        assert np.abs(frames[0] - frames[1]).max() < 1e-9
    else:  # the fast copy of this scene's variant adds fixed-point sums (quantum E 2^-52, E = 16; rt_device_types.h: sum_scale)
        assert np.abs(frames[0] ** 2 - frames[1] ** 2).max() < 1e-11  # in radiance: the quantum (3.6e-15) below the two arithmetics' own difference
        assert np.abs(frames[0] - frames[1]).max() < 5e-8
    assert np.abs(frames[1] - ref).max() < 1e-12    # unfused IEEE arithmetic: the oracle's own roundings, chunked sums aside


def band_rows(h, strip_rows, count, index):
    return ((np.arange(h) // strip_rows) % count) == index


def test_config2_at_full_size_and_full_spp_on_bands(rt, host, orc, gpu):
    """BASELINE config 2 exactly as benchmarked: three_balls 1920x1080, 256 spp, aperture 0.1."""
    s = host.Session(os.path.join(ROOT, "scenes", "config_c2.yml"), scene=os.path.join(ROOT, "scenes", "three_balls.yml"))
    p = s.params
    assert (p.width, p.height, p.samples, p.max_depth) == (1920, 1080, 256, 20)
    scene = rt.Scene(s)
    try:
        got = scene.render_frame(s.camera, p)
        stats = scene.last_stats()
    finally:
        scene.close()
    assert stats.samples == 1920 * 1080 * 256
    p.strip_rows, p.strip_count, p.strip_index = 4, 90, 51      # 3 bands of 4 rows: sky, spheres (row 564..567), ground
    ref, _ = orc.render(s.desc, s.camera, p)
    rows = band_rows(p.height, 4, 90, 51)
    d = np.abs(ref[rows] - got[rows])
    assert d.max() < TOL
    assert float((d.max(axis=-1) > TIGHT).mean()) < 2e-3
    assert got[rows].std() > 0.05


def test_config4_at_full_size_and_full_spp_on_bands(rt, host, orc, gpu):
    """BASELINE config 4 exactly as benchmarked: noise_and_textures 1920x1080, 512 spp (Perlin marble,
    earth image, checker ground, glass)."""
    s = host.Session(os.path.join(ROOT, "scenes", "config_c4.yml"), scene=os.path.join(ROOT, "scenes", "noise_and_textures.yml"))
    p = s.params
    assert (p.width, p.height, p.samples, p.max_depth) == (1920, 1080, 512, 20)
    scene = rt.Scene(s)
    try:
        got = scene.render_frame(s.camera, p)
        stats = scene.last_stats()
    finally:
        scene.close()
    assert stats.samples == 1920 * 1080 * 512
    p.strip_rows, p.strip_count, p.strip_index = 2, 180, 100    # 3 bands of 2 rows (rows 200, 560, 920)
    ref, _ = orc.render(s.desc, s.camera, p)
    rows = band_rows(p.height, 2, 180, 100)
    d = np.abs(ref[rows] - got[rows])
    assert d.max() < TOL
    assert float((d.max(axis=-1) > TIGHT).mean()) < 2e-3
    assert got[rows].std() > 0.05


def test_config5_share_at_full_spp(rt, orc, gpu):
    """BASELINE config 5 (cornell_box 3840x2160, 4096 spp, 8 GPUs): ONE rank's share exactly as that rank
    renders it — every 8th strip of 8 rows, 4096 spp, i.e. the chunk-length-64 path — against an oracle band
    of 2 of its rows through the lit box."""
    bundle, cam, _ = S.cornell_box()
    w, h, spp, rank = 3840, 2160, 4096, 3
    camera = S.camera_for(cam, w, h)
    scene = rt.Scene(bundle)
    try:
        part = scene.render_frame(camera, S.abi.render_params(w, h, spp, strip_rows=8, strip_count=8, strip_index=rank))
        stats = scene.last_stats()
    finally:
        scene.close()
    own = band_rows(h, 8, 8, rank)
    assert stats.samples == int(own.sum()) * w * spp
    lit = np.zeros(h, dtype=bool)
    lit[200:1960] = True                     # rows the open box covers at 16:9 (black background above and below)
    assert (part[~own] == 0).all() and (part[own & lit].max(axis=(1, 2)) > 0).all()
    band = S.abi.render_params(w, h, spp, strip_rows=2, strip_count=h // 2, strip_index=524)   # rows 1048, 1049 (strip 131 = rank 3)
    ref, _ = orc.render(bundle.desc, camera, band)
    rows = band_rows(h, 2, h // 2, 524)
    assert own[rows].all()
    d = np.abs(ref[rows] - part[rows])
    assert d.max() < TOL
    assert float((d.max(axis=-1) > TIGHT).mean()) < 2e-3
    assert part[rows].max() > 0.3


# ---------------------------------------------------------------- several devices, one process
def test_multi_device_call_on_one_card(rt, orc, gpu):
    """rt_render_frame_multi shards a frame over RtScene objects the way cpu.rs:118-131 shards it over
    threads.  On a one-GPU box the shares are several scenes on device 0 (own streams, own strips): the
    assembled frame must equal the single-scene frame bit for bit for every share count."""
    bundle, cam, _ = S.cornell_box_boxes()
    w, h, spp = 200, 131, 16          # 131 rows: a short last strip
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)
    scenes = [rt.Scene(bundle) for _ in range(3)]
    try:
        whole = scenes[0].render_frame(camera, params)
        for n, strip_rows in ((1, 0), (2, 0), (3, 8), (2, 16), (3, 5)):
            got = rt.render_frame_multi(scenes[:n], camera, params, strip_rows)
            if strip_rows % 8 == 0:   # strips made of whole 8-row item tiles: the very same sums
                assert np.array_equal(got, whole), (n, strip_rows)
            else:                     # other strip heights regroup pixels into other item tiles: equal to rounding
                assert np.abs(got - whole).max() < 1e-12, (n, strip_rows)
            traced = sum(int(s.last_stats().samples) for s in scenes[:n])
            assert traced == w * h * spp, (n, strip_rows)
        ref, _ = orc.render(bundle.desc, camera, params, use_bvh=0)
        assert np.abs(whole - ref).max() < TOL
        # argument errors
        with pytest.raises(rt.RtError) as e:
            rt.render_frame_multi([scenes[0], scenes[0]], camera, params)
        assert e.value.code == S.abi.RT_ERR_INVALID_ARGUMENT
        with pytest.raises(rt.RtError) as e:
            rt.render_frame_multi(scenes[:2], camera, S.abi.render_params(w, h, spp, strip_rows=8, strip_count=2, strip_index=0))
        assert e.value.code == S.abi.RT_ERR_INVALID_ARGUMENT
    finally:
        for s in scenes:
            s.close()


def test_multi_device_call_into_device_memory(rt, gpu):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU (the library does): no way to allocate the device buffer for this test")
    bundle, cam, _ = S.three_balls()
    w, h, spp = 160, 90, 8
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)
    scenes = [rt.Scene(bundle) for _ in range(2)]
    try:
        whole = scenes[0].render_frame(camera, params)
        out = torch.full((h, w, 3), -1.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        rt.render_frame_multi_device(scenes, camera, params, out.data_ptr())
        assert np.array_equal(out.cpu().numpy(), whole)
    finally:
        for s in scenes:
            s.close()


def test_staged_gather_runs_the_peer_copy_branch_on_one_card(rt, gpu):
    """rt_render_frame_multi_device with RtSceneOptions.gather = RT_GATHER_STAGED: every share renders into its OWN
    frame and its strips travel by the strided copy (+ the copy of a short last strip) that several devices use —
    csrc/rt_multi.hip step 2, which a one-GPU box never reaches otherwise because a share on the output's device renders
    in place.  2 and 3 shares, strips of 8 and 5 rows, 131 rows (a short last strip), one staged share alone, a mix
    of staged and in-place shares: the output must equal the single-scene frame (cpu.rs:118-131 shards ONE frame)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU (the library does): no way to allocate the device buffer for this test")
    bundle, cam, _ = S.cornell_box_boxes()
    w, h, spp = 200, 131, 16
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)
    staged = [rt.Scene(bundle, gather=S.abi.RT_GATHER_STAGED) for _ in range(3)]
    plain = [rt.Scene(bundle) for _ in range(3)]

    def multi_device(scenes, strip_rows):
        out = torch.full((h, w, 3), -1.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        rt.render_frame_multi_device(scenes, camera, params, out.data_ptr(), strip_rows)
        return out.cpu().numpy()

    try:
        whole = plain[0].render_frame(camera, params)
        for n, strip_rows in ((1, 0), (2, 0), (3, 8), (2, 16), (3, 5), (2, 5), (3, 24)):
            got = multi_device(staged[:n], strip_rows)
            assert (got != -1.0).all(), (n, strip_rows)               # every row arrived
            in_place = multi_device(plain[:n], strip_rows)
            assert np.array_equal(got, in_place), (n, strip_rows)     # same kernel work, another road to the output
            if strip_rows % 8 == 0:
                assert np.array_equal(got, whole), (n, strip_rows)
            else:
                assert np.abs(got - whole).max() < 1e-12, (n, strip_rows)
        mixed = multi_device([plain[0], staged[1], staged[2]], 8)      # share 0 in place, 1 and 2 copied
        assert np.array_equal(mixed, whole)
        # more shares than strips: shares 2.. own nothing (h = 131, strips of 64 rows: 3 strips for 3 shares, of 128: 2)
        assert np.array_equal(multi_device(staged, 128), whole)
    finally:
        for s in staged + plain:
            s.close()


def test_device_counted_samples_follow_the_work(rt, gpu):
    """RtRenderStats.samples is counted on the device, where a path is handed out: strips, column windows
    of the tile stream and the preview grid all report what they really traced."""
    bundle, cam, _ = S.two_balls()
    w, h, spp = 100, 60, 5
    camera = S.camera_for(cam, w, h)
    scene = rt.Scene(bundle)
    try:
        scene.render_frame(camera, S.abi.render_params(w, h, spp))
        assert scene.last_stats().samples == w * h * spp
        scene.render_frame(camera, S.abi.render_params(w, h, spp, strip_rows=8, strip_count=3, strip_index=1))
        own = int((((np.arange(h) // 8) % 3) == 1).sum())
        assert scene.last_stats().samples == own * w * spp
        scene.render_tiles(camera, S.abi.render_params(w, h, spp, tiles_w=7, tiles_h=3))
        assert scene.last_stats().samples == w * h * spp
        scene.render_frame(camera, S.abi.render_params(w, h, spp, tiles_w=2, tiles_h=2, scale=4))
        # cpu_scaled.rs:18-41: step = largest divisor <= scale of the tile size: 50 -> 2, 30 -> 3
        assert scene.last_stats().samples == (w // 2) * (h // 3) * spp
    finally:
        scene.close()
