"""GPU edge cases of the render path: degenerate scenes and frame shapes, sample
counts around the chunk boundaries, bad arguments, and the per-GPU share of
BASELINE config 5 (cornell_box 3840x2160, 8 GPUs) on one card."""
import numpy as np
import pytest

import scenes_py as S

pytestmark = pytest.mark.gpu

TOL = 1e-3
TIGHT = 1e-9


def _parity(rt, orc, bundle, cam, w, h, spp, tm="None", **kw):
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp, **kw)
    ref, ref_segs = orc.render(bundle.desc, camera, params)
    scene = rt.Scene(bundle)
    try:
        got = scene.render_frame(camera, params)
        stats = scene.last_stats()
    finally:
        scene.close()
    d = np.abs(got - ref)
    assert np.isfinite(got).all()
    assert d.max() < TOL
    assert (d > TIGHT).mean() < 2e-3
    return got, ref, stats, ref_segs


def test_empty_scene_is_the_background(rt, orc, gpu):
    """No primitives at all: every ray misses (renderer.rs:79-89) - Sky gradient or the
    solid colour; one segment per sample."""
    cam = dict(look_from=(0.0, 0.0, 0.0), look_at=(0.0, 0.0, -1.0), vfov=40.0, aperture=0.0, focus_distance=1.0)
    for bg in (S.abi.sky(), S.abi.solid_background((0.25, 0.5, 0.75))):
        bundle = S.abi.SceneBundle([], [], [], bg)
        got, ref, stats, ref_segs = _parity(rt, orc, bundle, cam, 40, 24, 3)
        assert stats.segments == 40 * 24 * 3 == ref_segs
    assert np.allclose(got, np.sqrt([0.25, 0.5, 0.75]))  # solid: sqrt(sum / samples) of a constant


@pytest.mark.parametrize("w,h", [(2, 2), (9, 2), (2, 9), (3, 64)])
def test_thin_frames(rt, orc, gpu, w, h):
    """Two pixels is the smallest regular dimension (u, v in {0 + j, 1 + j} / 1)."""
    bundle, cam, _ = S.three_balls()
    _parity(rt, orc, bundle, cam, w, h, 4)


@pytest.mark.parametrize("w,h", [(1, 1), (1, 8), (8, 1)])
def test_one_pixel_dimension_is_refused(rt, gpu, w, h):
    """cpu.rs:36,40 divide by (W-1) and (H-1): with a one-pixel dimension the reference
    divides by zero (inf/NaN rays).  The device path refuses the frame instead."""
    bundle, cam, _ = S.two_balls()
    scene = rt.Scene(bundle)
    try:
        with pytest.raises(rt.RtError) as e:
            scene.render_frame(S.camera_for(cam, 8, 8), S.abi.render_params(w, h, 1))
        assert e.value.code == S.abi.RT_ERR_INVALID_ARGUMENT
    finally:
        scene.close()


@pytest.mark.parametrize("spp", [1, 7, 8, 9, 31, 32, 33, 47, 48, 49, 64, 65, 200, 257, 1000, 2049])
def test_sample_counts_around_chunk_boundaries(rt, orc, gpu, spp):
    """The pooled kernel sums a pixel's samples in chunks: full-length ones of spp / 16 (at least 24) samples,
    then a taper of halving chunks down to 8 or fewer (rt_api.hip: chunk_plan); any count must give the
    oracle's frame."""
    bundle, cam, _ = S.cornell_box()
    w, h = (16, 9) if spp > 100 else (48, 27)
    _parity(rt, orc, bundle, cam, w, h, spp)


def test_invalid_arguments_are_errors_not_faults(rt, gpu):
    bundle, cam, _ = S.two_balls()
    camera = S.camera_for(cam, 16, 9)
    scene = rt.Scene(bundle)
    try:
        for bad in (dict(width=1), dict(height=1), dict(samples=0), dict(max_depth=-1), dict(max_depth=1 << 24),
                    dict(strip_count=2, strip_rows=0), dict(strip_count=2, strip_rows=8, strip_index=2),
                    dict(scale=-1)):
            p = S.abi.render_params(16, 9, 2)
            for k, v in bad.items():
                setattr(p, k, v)
            with pytest.raises(rt.RtError) as e:
                scene.render_frame(camera, p)
            assert e.value.code == S.abi.RT_ERR_INVALID_ARGUMENT, bad
        # and the scene still renders afterwards
        assert np.isfinite(scene.render_frame(camera, S.abi.render_params(16, 9, 2))).all()
    finally:
        scene.close()


def test_config5_share_of_one_gpu(rt, orc, gpu):
    """BASELINE config 5 is cornell_box 3840x2160 on 8 GPUs.  One rank's share (strips of 8
    rows, every 8th strip) at reduced spp on this card: rows it owns equal the whole-frame
    render bit for bit, rows it does not own are untouched, the oracle agrees on them."""
    bundle, cam, _ = S.cornell_box()
    w, h, spp = 3840, 2160, 2
    camera = S.camera_for(cam, w, h)
    scene = rt.Scene(bundle)
    try:
        full = scene.render_frame(camera, S.abi.render_params(w, h, spp))
        for rank in (0, 5):
            p = S.abi.render_params(w, h, spp, strip_rows=8, strip_count=8, strip_index=rank)
            part = scene.render_frame(camera, p)
            st = scene.last_stats()
            own = ((np.arange(h) // 8) % 8) == rank
            assert np.array_equal(part[own], full[own])
            assert (part[~own] == 0).all()
            assert st.samples == int(own.sum()) * w * spp
        band = S.abi.render_params(w, h, spp, strip_rows=8, strip_count=270, strip_index=130)   # 8 rows through the box
        ref, _ = orc.render(bundle.desc, camera, band)
        rows = ((np.arange(h) // 8) % 270) == 130
        assert np.abs(ref[rows] - full[rows]).max() < TOL
    finally:
        scene.close()


def _noise_scene(shuffled):
    """A marble sphere on a checker ground: Noise texture over a Perlin table whose
    permutation tables are either the identity (what the reference always has:
    noise.rs:121-130 never shuffles) or a real permutation (the ABI allows it)."""
    abi = S.abi
    rng = np.random.default_rng(7)
    pl = abi.RtPerlin()
    g = rng.uniform(-1.0, 1.0, size=(256, 3))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    for i in range(256):
        for k in range(3):
            pl.ranvec[i][k] = float(g[i, k])
    perms = [rng.permutation(256) if shuffled else np.arange(256) for _ in range(3)]
    for i in range(256):
        pl.perm_x[i], pl.perm_y[i], pl.perm_z[i] = int(perms[0][i]), int(perms[1][i]), int(perms[2][i])
    noise = abi.RtTexture(abi.RT_TEX_NOISE, 0, 0, 0, 0, 7, abi.D3(1.0, 1.0, 1.0), 4.0)
    checker = abi.RtTexture(abi.RT_TEX_CHECKERED, 2, 3, 0, 0, 0, abi.D3(0, 0, 0), 0.0)
    textures = [noise, checker, abi.solid((0.2, 0.3, 0.1)), abi.solid((0.9, 0.9, 0.9))]
    materials = [abi.material(abi.RT_MAT_LAMBERTIAN, 0), abi.material(abi.RT_MAT_LAMBERTIAN, 1)]
    prims = [abi.sphere((0.0, 2.0, 0.0), 2.0, 0, 1), abi.sphere((0.0, -1000.0, 0.0), 1000.0, 1, 2)]
    cam = dict(look_from=(13.0, 2.0, 3.0), look_at=(0.0, 0.0, 0.0), vfov=20.0, aperture=0.0, focus_distance=10.0)
    return abi.SceneBundle(prims, materials, textures, abi.sky(), perlins=[pl]), cam


@pytest.mark.parametrize("shuffled", [False, True])
def test_noise_with_identity_and_shuffled_permutations(rt, orc, gpu, shuffled):
    """The device hashes the lattice as (i ^ j ^ k) & 255 when every permutation table is the
    identity and reads the tables otherwise; both must give the oracle's marble."""
    bundle, cam = _noise_scene(shuffled)
    got, ref, _, _ = _parity(rt, orc, bundle, cam, 96, 54, 6)
    assert got.std() > 0.05


def test_two_scenes_from_two_threads(rt, orc, gpu):
    """rt_abi.h threading contract: different RtScene objects may be used from different threads
    at once (ctypes drops the GIL during the calls); each must get the frame it gets alone."""
    import threading
    jobs = []
    for scene_fn, (w, h, spp) in ((S.cornell_box, (160, 90, 24)), (S.three_balls, (128, 72, 16))):
        bundle, cam, _ = scene_fn()
        jobs.append((bundle, S.camera_for(cam, w, h), S.abi.render_params(w, h, spp)))
    alone = []
    for bundle, camera, params in jobs:
        scene = rt.Scene(bundle)
        try:
            alone.append(scene.render_frame(camera, params))
        finally:
            scene.close()
    together = [None] * len(jobs)
    errors = []

    def work(k):
        try:
            bundle, camera, params = jobs[k]
            scene = rt.Scene(bundle)
            try:
                for _ in range(3):
                    together[k] = scene.render_frame(camera, params)
            finally:
                scene.close()
        except Exception as e:  # noqa: BLE001 - reported below
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for a, b in zip(alone, together):
        assert np.array_equal(a, b)


def test_furnace_known_answer_without_the_oracle(rt, gpu):
    """An answer derived by hand, not by the oracle: a convex Lambertian object of albedo a in a
    uniform environment of radiance c.  Every path that hits it scatters once and escapes
    (a convex body cannot be hit again from its own surface), so EVERY sample returns a * c
    exactly and every covered pixel is sqrt(a * c) (cpu.rs:52 gamma); the rest see sqrt(c).
    lambertian.rs:26-38 + renderer.rs:41-90 + background_color.rs:45-48."""
    abi = S.abi
    a = np.array([0.5, 0.25, 0.75])
    c = np.array([0.8, 0.6, 0.4])
    bundle = abi.SceneBundle([abi.sphere((0.0, 0.0, -3.0), 1.0, 0)], [abi.material(abi.RT_MAT_LAMBERTIAN, 0)],
                             [abi.solid(tuple(a))], abi.solid_background(tuple(c)))
    cam = dict(look_from=(0.0, 0.0, 0.0), look_at=(0.0, 0.0, -1.0), vfov=60.0, aperture=0.0, focus_distance=1.0)
    w, h, spp = 128, 72, 37
    scene = rt.Scene(bundle)
    try:
        got = scene.render_frame(S.camera_for(cam, w, h), abi.render_params(w, h, spp))
        stats = scene.last_stats()
    finally:
        scene.close()
    on = np.isclose(got, np.sqrt(a * c), rtol=1e-13, atol=0).all(axis=-1)
    off = np.isclose(got, np.sqrt(c), rtol=1e-13, atol=0).all(axis=-1)
    edge = ~(on | off)                      # pixels whose samples straddle the silhouette
    assert on.sum() > 600 and off.sum() > 5000 and edge.sum() < 300
    lo, hi = np.minimum(np.sqrt(a * c), np.sqrt(c)), np.maximum(np.sqrt(a * c), np.sqrt(c))
    assert ((got[edge] >= lo - 1e-12) & (got[edge] <= hi + 1e-12)).all()
    # one segment for a miss, two for a hit: between 1 and 2 segments per sample, and an integer total
    assert stats.samples == w * h * spp and stats.samples < stats.segments < 2 * stats.samples


def test_white_furnace_is_exactly_white(rt, gpu):
    """White furnace, again without the oracle: albedo-1 Lambertian, fuzz-0 and fuzzy Metal of
    albedo 1, glass (attenuation 1, dialectric.rs:28) and a unit light in a white environment.
    Every path ends in the environment (1), in the light (1) or in depth exhaustion (white,
    renderer.rs:48-55), with throughput exactly 1 - unless a fuzzy reflection dips below the
    surface and is absorbed (metal.rs:38-42, black).  So every pixel is exactly 1, except
    pixels that see the fuzzy sphere, which may be darker but never brighter."""
    abi = S.abi
    one = abi.solid((1.0, 1.0, 1.0))
    L, M, D, E = abi.RT_MAT_LAMBERTIAN, abi.RT_MAT_METAL, abi.RT_MAT_DIELECTRIC, abi.RT_MAT_DIFFUSE_LIGHT
    materials = [abi.material(L, 0), abi.material(M, 0, fuzz=0.0), abi.material(D, -1, ior=1.5), abi.material(E, 0)]
    prims = [abi.sphere((0.0, -100.5, -1.0), 100.0, 0), abi.sphere((-1.1, 0.0, -1.0), 0.5, 1),
             abi.sphere((0.0, 0.0, -1.0), 0.5, 2), abi.sphere((0.0, 0.0, -1.0), -0.4, 2),
             abi.sphere((1.1, 0.0, -1.0), 0.5, 3)]
    cam = dict(look_from=(0.0, 1.0, 4.0), look_at=(0.0, 0.0, -1.0), vfov=35.0, aperture=0.2, focus_distance=5.0)
    w, h, spp = 160, 90, 24
    for fuzzy in (False, True):
        mats = list(materials)
        if fuzzy:
            mats[1] = abi.material(M, 0, fuzz=0.6)
        bundle = abi.SceneBundle(prims, mats, [one], abi.solid_background((1.0, 1.0, 1.0)))
        scene = rt.Scene(bundle)
        try:
            got = scene.render_frame(S.camera_for(cam, w, h), abi.render_params(w, h, spp, max_depth=12))
        finally:
            scene.close()
        assert np.isfinite(got).all()
        if not fuzzy:
            assert (got == 1.0).all()
        else:
            assert (got <= 1.0).all() and (got == 1.0).mean() > 0.7 and got.min() > 0.3


@pytest.mark.parametrize("scene_fn", [S.three_balls, S.cornell_box, S.cornell_box_boxes])
def test_v1_kernel_still_matches_the_oracle(rt, orc, gpu, scene_fn):
    """RtSceneOptions.kernel = RT_KERNEL_V1 selects the lane-per-pixel kernel, the second implementation
    of the same contract: it must agree with the oracle, and with the pooled kernel to rounding."""
    bundle, cam, _ = scene_fn()
    w, h, spp = 96, 54, 12
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)
    ref, ref_segs = orc.render(bundle.desc, camera, params, use_bvh=0)
    frames = {}
    for name in ("pool", "v1"):
        scene = rt.Scene(bundle, kernel=S.abi.RT_KERNEL_V1 if name == "v1" else S.abi.RT_KERNEL_POOL)
        try:
            frames[name] = scene.render_frame(camera, params)
            segs = scene.last_stats().segments
        finally:
            scene.close()
        d = np.abs(frames[name] - ref)
        assert d.max() < TOL and (d > TIGHT).mean() < 2e-3, name
        assert abs(int(segs) - ref_segs) <= 4
    if scene_fn is not S.cornell_box_boxes:
        assert np.abs(frames["pool"] - frames["v1"]).max() < 1e-9
    else:
        # The pooled variants that keep two items in flight (any primitive kind, BVH) add FIXED-POINT sums: an absolute
        # quantum of E 2^-52 per sample (E = 16 bounds this scene's radiance; rt_device_types.h: sum_scale) where a double
        # has a relative one.  In radiance — the square of what the frame holds — the two kernels agree to that quantum; a
        # pixel that is black for every purpose may come out up to sqrt(E 2^-53) = 4e-8 apart.
        assert np.abs(frames["pool"] ** 2 - frames["v1"] ** 2).max() < 1e-13  # (the quantum, 3.6e-15, and the order of the f64 sums)
        assert np.abs(frames["pool"] - frames["v1"]).max() < 5e-8


def test_tile_stream_refuses_strips(rt, gpu):
    bundle, cam, _ = S.two_balls()
    scene = rt.Scene(bundle)
    try:
        p = S.abi.render_params(32, 18, 2, tiles_w=2, tiles_h=2, strip_rows=8, strip_count=2, strip_index=0)
        with pytest.raises(rt.RtError) as e:
            scene.render_tiles(S.camera_for(cam, 32, 18), p)
        assert e.value.code == S.abi.RT_ERR_INVALID_ARGUMENT
    finally:
        scene.close()


@pytest.mark.parametrize("n", [3000, 20000])
def test_many_primitives_through_the_global_memory_bvh(rt, orc, gpu, n):
    """Thousands of spheres: the BVH node array no longer fits in LDS, so the walk reads it from
    global memory; the oracle's own BVH (a different tree) must see the same closest hits.
    A hall of small mirrors and marbles is also chaotic: a bounce off a sphere of radius r at
    distance D multiplies a direction error by ~D/r, so the 1-2 ulp by which the device's
    reciprocal-based divisions differ from true divisions reach 1e-7 after a few
    bounces at 3 000 spheres and flip the odd hit at 20 000.  The same paths are traced (segment
    counts agree to a few in 70 000); the per-pixel bound is checked where the scene lets it hold.
    This is the STATISTICAL check of the product build.  That the outliers are arithmetic and not a traversal
    defect is proven in tests/test_gpu_parity_proofs.py: the same spheres with max_depth 1 and one colour each
    hold the full tolerance on every pixel, and so does this very scene through the exact-arithmetic build."""
    abi = S.abi
    rng = np.random.default_rng(5)
    centers = rng.uniform(-40.0, 40.0, size=(n, 3))
    centers[:, 2] = rng.uniform(-90.0, -10.0, size=n)
    radii = rng.uniform(0.2, 0.9, size=n)
    textures = [abi.solid((0.8, 0.3, 0.3)), abi.solid((0.3, 0.8, 0.3)), abi.solid((0.9, 0.9, 0.9))]
    materials = [abi.material(abi.RT_MAT_LAMBERTIAN, 0), abi.material(abi.RT_MAT_LAMBERTIAN, 1),
                 abi.material(abi.RT_MAT_METAL, 2, fuzz=0.1), abi.material(abi.RT_MAT_DIELECTRIC, -1, ior=1.5)]
    prims = [abi.sphere(tuple(centers[i]), float(radii[i]), int(i % 4), i) for i in range(n)]
    bundle = abi.SceneBundle(prims, materials, textures, abi.sky())
    cam = dict(look_from=(0.0, 0.0, 5.0), look_at=(0.0, 0.0, -50.0), vfov=50.0, aperture=0.0, focus_distance=10.0)
    w, h, spp = 96, 54, 3
    camera = S.camera_for(cam, w, h)
    params = abi.render_params(w, h, spp, max_depth=8)
    ref, ref_segs = orc.render(bundle.desc, camera, params)
    scene = rt.Scene(bundle)
    try:
        got = scene.render_frame(camera, params)
        stats = scene.last_stats()
    finally:
        scene.close()
    d = np.abs(got - ref)
    assert np.isfinite(got).all() and got.std() > 0.05
    assert abs(int(stats.segments) - ref_segs) <= 8          # the same paths
    assert np.median(d) < 1e-13 and d.mean() < 1e-4
    if n <= 3000:
        assert d.max() < TOL and (d > TIGHT).mean() < 5e-3
    else:
        assert (d > TOL).mean() < 0.02   # RT_ARITH_FAST: 1-2 ulp amplified by the hall of mirrors flip the odd hit ...
    # ... and a caller who needs the north-star tolerance on such a scene asks for the reference's arithmetic
    # (RtSceneOptions.arithmetic): full tolerance on EVERY pixel, the oracle's exact segment count
    scene = rt.Scene(bundle, arithmetic=abi.RT_ARITH_REFERENCE)
    try:
        got = scene.render_frame(camera, params)
        stats = scene.last_stats()
    finally:
        scene.close()
    d = np.abs(got - ref)
    assert int(stats.segments) == ref_segs
    assert d.max() < TOL and (d > TIGHT).mean() < 1e-3
