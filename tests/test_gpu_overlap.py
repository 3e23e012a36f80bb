"""The pooled variants that keep TWO items in flight per wave (any primitive kind, BVH: rt_trace_pool_kernel.hip, OVERLAP)
and their fixed-point sums (rt_device_types.h: sum_scale).

What has to hold: a finished item's sums are integers, so they do not depend on which paths a wave traced side by side —
the frame is bit-identical for every strip layout, tile-aligned or not, for every launch shape and from run to run; the
sums agree with the f64 sums of the RT_ARITH_REFERENCE copy to the quantum; a scene whose radiance has no bound (a colour
above 1 on a scattering material) is rendered by the RT_ARITH_REFERENCE copy instead; chunks longer than 2048 samples
halve the scale rather than overflow the 64-bit sums.
"""
import numpy as np
import pytest

import scenes_py as S

pytestmark = pytest.mark.gpu
TOL = 1e-3


def strips(scene, camera, w, h, spp, rows, count):
    out = np.zeros((h, w, 3))
    for index in range(count):
        part = scene.render_frame(camera, S.abi.render_params(w, h, spp, strip_rows=rows, strip_count=count, strip_index=index))
        own = ((np.arange(h) // rows) % count) == index
        out[own] = part[own]
    return out


def test_frames_do_not_depend_on_strips_that_cut_tiles(rt, gpu):
    """Strips of 5 or 3 rows cut the 8x8 item tiles: every item then holds other pixels, starts and ends beside other
    items — and, with integer sums, the frame is the same to the last bit (the one-item-at-a-time variants, whose f64
    sums depend on the order their samples finish in, agree to 1e-12 here: tests/test_gpu_parity_proofs.py)."""
    bundle, cam, _ = S.cornell_box_boxes()
    w, h, spp = 160, 90, 24
    camera = S.camera_for(cam, w, h)
    scene = rt.Scene(bundle)
    try:
        whole = scene.render_frame(camera, S.abi.render_params(w, h, spp))
        assert np.array_equal(scene.render_frame(camera, S.abi.render_params(w, h, spp)), whole)   # run to run
        for rows, count in ((5, 3), (3, 2), (8, 4), (7, 5)):
            assert np.array_equal(strips(scene, camera, w, h, spp, rows, count), whole), (rows, count)
        # another tile grid of the stream, another region order of the delivering launch: the same pixels
        stitched = np.zeros_like(whole)
        for r, c, tw, th, t in scene.render_tiles(camera, S.abi.render_params(w, h, spp, tiles_w=7, tiles_h=3)):
            stitched[r:r + th, c:c + tw] = t
        assert np.array_equal(stitched, whole)
    finally:
        scene.close()


def test_fixed_point_sums_agree_with_f64_sums_to_the_quantum(rt, orc, gpu):
    """Fast against reference arithmetic differs in the divisions' last bits AND in the sums; in radiance (the square of
    what the frame holds) the quantum E 2^-52 = 3.6e-15 per sample stays below the arithmetic's own difference."""
    bundle, cam, _ = S.cornell_box_boxes()
    w, h, spp = 128, 72, 32
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)
    frames = {}
    for name, arithmetic in (("fast", S.abi.RT_ARITH_FAST), ("reference", S.abi.RT_ARITH_REFERENCE)):
        scene = rt.Scene(bundle, arithmetic=arithmetic)
        try:
            frames[name] = scene.render_frame(camera, params)
        finally:
            scene.close()
    ref, _ = orc.render(bundle.desc, camera, params, use_bvh=0)
    assert np.abs(frames["fast"] ** 2 - frames["reference"] ** 2).max() < 1e-11
    assert np.abs(frames["fast"] - ref).max() < TOL and np.abs(frames["reference"] - ref).max() < 1e-11
    # where the picture is not black the gamma-encoded values agree as closely as before
    lit = ref > 1e-3
    assert np.abs(frames["fast"] - frames["reference"])[lit].max() < 1e-10


def unbounded(bundle):
    """The same scene with a colour above 1 on a scattering material: its radiance has no bound."""
    textures = list(bundle.textures)
    textures[2] = S.abi.solid((1.25, 0.63, 0.63))
    return S.abi.SceneBundle(list(bundle.primitives), list(bundle.materials), textures, S.abi.solid_background((0.0, 0.0, 0.0)))


def test_a_scene_without_a_radiance_bound_takes_the_f64_sums(rt, orc, gpu):
    """A Lambertian colour of 1.25 lets a path's throughput grow: no scale fits every sample, so the two-items-in-flight
    variants hand the scene to their RT_ARITH_REFERENCE copies — whose frame it therefore equals bit for bit."""
    bundle, cam, _ = S.cornell_box_boxes()
    hot = unbounded(bundle)
    w, h, spp = 96, 54, 16
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)
    frames = []
    for arithmetic in (S.abi.RT_ARITH_FAST, S.abi.RT_ARITH_REFERENCE):
        scene = rt.Scene(hot, arithmetic=arithmetic)
        try:
            frames.append(scene.render_frame(camera, params))
        finally:
            scene.close()
    assert np.array_equal(frames[0], frames[1])
    ref, _ = orc.render(hot.desc, camera, params, use_bvh=0)
    assert np.abs(frames[0] - ref).max() < 1e-9
    assert ref.max() > 1.0 and np.isfinite(frames[0]).all()


def test_chunks_longer_than_2048_samples_halve_the_scale(rt, gpu):
    """36 864 samples per pixel make chunks of 2 304: 2 304 samples of up to 2^52 each would pass 2^63, so the host halves
    the scale.  The frame has to agree with the f64 sums of the reference copy as it does at 32 samples per pixel."""
    bundle, cam, _ = S.cornell_box_boxes()
    w, h, spp = 16, 16, 36864
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)
    frames = []
    for arithmetic in (S.abi.RT_ARITH_FAST, S.abi.RT_ARITH_REFERENCE):
        scene = rt.Scene(bundle, arithmetic=arithmetic)
        try:
            frames.append(scene.render_frame(camera, params))
            assert scene.last_stats().samples == w * h * spp
        finally:
            scene.close()
    assert np.abs(frames[0] ** 2 - frames[1] ** 2).max() < 1e-10
    assert np.abs(frames[0] - frames[1]).max() < 1e-7


def test_bvh_variant_frames_do_not_depend_on_strips_that_cut_tiles(rt, host, gpu):
    """The same for the BVH variant (485 spheres, moving ones among them: ray times in dynamic LDS, lens samples, one batch
    buffer): strips of 5 and 3 rows against the whole frame, bit for bit."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    session = host.Session(os.path.join(root, "scenes", "config_c2.yml"), scene="random")
    p = session.params
    w, h, spp = 160, 88, 6
    camera = host.camera_new((0, 2, 10), (0, 0, 0), 20.0, 0.1, 10.0, w, h)
    scene = rt.Scene(session)
    try:
        def frame(**strip):
            q = S.abi.render_params(w, h, spp, **strip)
            q.max_depth, q.seed = p.max_depth, p.seed
            return scene.render_frame(camera, q)
        whole = frame()
        assert np.isfinite(whole).all() and whole.max() > 0.1
        for rows, count in ((5, 3), (3, 4), (8, 2)):
            got = np.zeros_like(whole)
            for index in range(count):
                part = frame(strip_rows=rows, strip_count=count, strip_index=index)
                own = ((np.arange(h) // rows) % count) == index
                got[own] = part[own]
            assert np.array_equal(got, whole), (rows, count)
    finally:
        scene.close()


def test_default_scene_at_full_size_on_bands(rt, host, orc, gpu):
    """The reference's default content (the Cornell box WITH its two rotated boxes, scene/sandbox.rs:39-80) at 1920x1080 x
    128 spp through the two-item variant, against the oracle on three bands of rows: the fixed-point sums at the size and
    sample count that are benchmarked.  In radiance the frames agree to the quantum and the arithmetic's own difference."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = host.Session(os.path.join(root, "scenes", "config_c3.yml"), scene=os.path.join(root, "scenes", "cornell_box_boxes.yml"))
    p = s.params
    p.samples = 128
    assert (p.width, p.height, p.max_depth) == (1920, 1080, 20)
    scene = rt.Scene(s)
    try:
        got = scene.render_frame(s.camera, p)
        assert scene.last_stats().samples == 1920 * 1080 * 128
    finally:
        scene.close()
    p.strip_rows, p.strip_count, p.strip_index = 2, 180, 100    # rows 200-201, 560-561, 920-921
    # (the oracle's linear scan: through its BVH it replicates RotateY::create_bounding_box's arithmetic slip
    # (rotate_y.rs:77,83-84), which culls rays that do hit the rotated boxes — 169 of these 11 520 pixels)
    ref, _ = orc.render(s.desc, s.camera, p, use_bvh=0)
    rows = ((np.arange(p.height) // 2) % 180) == 100
    d = np.abs(ref[rows] - got[rows])
    assert d.max() < TOL
    assert np.abs(ref[rows] ** 2 - got[rows] ** 2).max() < 1e-9                 # radiance
    assert float((d.max(axis=-1) > 1e-9).mean()) < 2e-3                          # and almost everywhere in the frame's own terms
    assert got[rows].std() > 0.05


def with_light(bundle, emission):
    textures = list(bundle.textures)
    textures[3] = S.abi.solid((emission, emission, emission))
    return S.abi.SceneBundle(list(bundle.primitives), list(bundle.materials), textures, S.abi.solid_background((0.0, 0.0, 0.0)))


@pytest.mark.parametrize("emission", [16.0, 15.99999999, 1.0, 1e13])
def test_radiance_bounds_at_the_edges(rt, gpu, emission):
    """The scale of the fixed-point sums comes from the scene's largest emission: a bound that IS a power of two, one a hair
    below it (the exponent keeps a margin), the smallest one (1: the white of an exhausted depth), and one beyond 2^40 —
    where the scene takes the f64 sums of the reference copy.  Against that copy the radiance agrees to the quantum."""
    bundle, cam, _ = S.cornell_box_boxes()
    lit = with_light(bundle, emission)
    w, h, spp = 96, 54, 16
    camera = S.camera_for(cam, w, h)
    params = S.abi.render_params(w, h, spp)
    frames = []
    for arithmetic in (S.abi.RT_ARITH_FAST, S.abi.RT_ARITH_REFERENCE):
        scene = rt.Scene(lit, arithmetic=arithmetic)
        try:
            frames.append(scene.render_frame(camera, params))
        finally:
            scene.close()
    assert np.isfinite(frames[0]).all() and frames[0].max() > 0.0
    if emission > 2.0 ** 40:
        assert np.array_equal(frames[0], frames[1])
    else:
        assert np.abs(frames[0] ** 2 - frames[1] ** 2).max() < 1e-11 * max(1.0, emission)
