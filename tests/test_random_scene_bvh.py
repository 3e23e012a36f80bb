"""MovingSphere (geometry/moving_sphere.rs), the procedural `random` scene
(scene/random.rs) and the BVH that large scenes go through on the device."""
import ctypes as C
import importlib
import math
import os

import numpy as np
import pytest

import scenes_py as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
abi = S.abi
INF = float("inf")


@pytest.fixture(scope="module")
def host():
    return importlib.import_module("racer-tracer_amd.host")


def random_session(host, seed=1):
    return host.Session(os.path.join(ROOT, "scenes", "config_c1.yml"), scene="random", seed=seed)


# ------------------------------------------------------------------ oracle KATs
def hit_at(orc, prim, o, d, time, t_min=0.001, t_max=INF):
    h = orc.OrcHit()
    ok = orc.lib().orc_hit_primitive_time(C.byref(prim), abi.D3(*o), abi.D3(*d), time, t_min, t_max, C.byref(h))
    return bool(ok), h


def test_moving_sphere_follows_ray_time(orc):
    """moving_sphere.rs:37-39: centre = pos_a + (time - time_a)/(time_b - time_a) * (pos_b - pos_a)."""
    ms = abi.moving_sphere((0, 0, -5), (0, 2, -5), 0.5, 0)
    for time, y in ((0.0, 0.0), (0.25, 0.5), (1.0, 2.0)):
        ok, h = hit_at(orc, ms, (0, y, 0), (0, 0, -1), time)
        assert ok and h.t == pytest.approx(4.5) and list(h.normal) == pytest.approx([0, 0, 1])
        assert list(h.point) == pytest.approx([0, y, -4.5])
    assert not hit_at(orc, ms, (0, 0, 0), (0, 0, -1), 1.0)[0]      # at time 1 the sphere has moved away
    ok, h = hit_at(orc, ms, (0, 0, 0), (0, 0, -1), 0.0)
    # moving_sphere.rs:76: uv from the hit POINT, not the unit normal (SURVEY B-19):
    # point = (0,0,-4.5) -> theta = acos(-0) = pi/2, phi = atan2(4.5, 0) + pi = 3pi/2
    assert (h.u, h.v) == pytest.approx((0.75, 0.5))
    mn, mx = abi.D3(), abi.D3()
    orc.lib().orc_primitive_aabb(C.byref(ms), mn, mx)              # moving_sphere.rs:93-106
    assert list(mn) == [-0.5, -0.5, -5.5] and list(mx) == [0.5, 2.5, -4.5]


def test_moving_sphere_blurs_in_a_render(orc):
    """A fast vertical sphere smears over its travel: rows it only passes through are partly covered."""
    tex = [abi.solid((0.9, 0.1, 0.1))]
    mats = [abi.material(S.L, 0)]
    bundle = abi.SceneBundle([abi.moving_sphere((0, -1, -4), (0, 1, -4), 0.5, 0)], mats, tex, abi.sky())
    cam = S.camera_for(dict(look_from=(0, 0, 0), look_at=(0, 0, -1), vfov=60.0, aperture=0.0, focus_distance=1.0), 32, 32)
    frame, _ = orc.render(bundle.desc, cam, abi.render_params(32, 32, 64))
    blue = frame[:, 16, 2]                       # the default sky has blue = 1 everywhere; the red sphere removes it
    covered = blue < 0.98
    assert covered.sum() >= 16                   # travel (2 units ~ 14 rows) + diameter (~7 rows); a static sphere covers ~7
    assert blue.min() > 0.5                      # never opaque: it is in front of any given row only part of the time
    static = abi.SceneBundle([abi.sphere((0, 0, -4), 0.5, 0)], mats, tex, abi.sky())
    still, _ = orc.render(static.desc, cam, abi.render_params(32, 32, 64))
    assert (still[:, 16, 2] < 0.98).sum() <= 9 and still[:, 16, 2].min() < 0.5


# --------------------------------------------------------------- random loader
def test_random_scene_structure(host):
    s = random_session(host)
    d = s.desc
    prims = [d.primitives[i] for i in range(d.n_primitives)]
    assert 440 <= len(prims) <= 488
    ground, glass, matte, metal = prims[0], prims[-3], prims[-2], prims[-1]
    assert (ground.kind, list(ground.p)[:4]) == (abi.RT_PRIM_SPHERE, [0.0, -1000.0, 0.0, 1000.0])
    assert d.textures[d.materials[ground.material].texture].kind == abi.RT_TEX_CHECKERED
    assert list(glass.p)[:4] == [0.0, 1.0, 0.0, 1.0] and d.materials[glass.material].kind == abi.RT_MAT_DIELECTRIC
    assert list(matte.p)[:4] == [-4.0, 1.0, 0.0, 1.0] and d.materials[matte.material].kind == abi.RT_MAT_LAMBERTIAN
    assert list(metal.p)[:4] == [4.0, 1.0, 0.0, 1.0] and d.materials[metal.material].fuzz == 0.0
    small = prims[1:-3]
    kinds = {abi.RT_MAT_LAMBERTIAN: 0, abi.RT_MAT_METAL: 0, abi.RT_MAT_DIELECTRIC: 0}
    for p in small:
        m = d.materials[p.material]
        kinds[m.kind] += 1
        assert p.p[1] == 0.2 and p.p[3] == 0.2
        assert math.hypot(p.p[0] - 4.0, p.p[2]) > 0.9                   # random.rs:46 keeps clear of the big metal ball
        assert -11.0 <= p.p[0] < 10.9 and -11.0 <= p.p[2] < 10.9
        if m.kind == abi.RT_MAT_LAMBERTIAN:                              # diffuse ones move (random.rs:50-54)
            assert p.kind == abi.RT_PRIM_MOVING_SPHERE and (p.time_a, p.time_b) == (0.0, 1.0)
            assert p.center_b[0] == p.p[0] and p.center_b[2] == p.p[2] and 0.2 <= p.center_b[1] < 0.7
        else:
            assert p.kind == abi.RT_PRIM_SPHERE
            if m.kind == abi.RT_MAT_METAL:
                assert 0.0 <= m.fuzz < 0.5 and all(0.5 <= c < 1.0 for c in d.textures[m.texture].color)
    n = len(small)
    assert 0.72 < kinds[abi.RT_MAT_LAMBERTIAN] / n < 0.88               # choose_mat < 0.8
    assert 0.08 < kinds[abi.RT_MAT_DIELECTRIC] / n < 0.22               # 0.8 .. 0.95
    assert 0.01 < kinds[abi.RT_MAT_METAL] / n < 0.10                    # > 0.95
    assert s.camera.vfov == 20.0 and s.camera.lens_radius == 0.05 and list(s.camera.origin) == [0.0, 2.0, 10.0]
    assert d.background.kind == abi.RT_BG_SKY


def test_random_scene_is_a_function_of_the_seed(host, orc):
    a, b, c = random_session(host, 1), random_session(host, 1), random_session(host, 2)
    pa = [tuple(a.desc.primitives[i].p) for i in range(a.desc.n_primitives)]
    pb = [tuple(b.desc.primitives[i].p) for i in range(b.desc.n_primitives)]
    pc = [tuple(c.desc.primitives[i].p) for i in range(c.desc.n_primitives)]
    assert pa == pb and pa != pc
    # the loader's n-th random_double() is draw (pixel=n, RT_RNG_SAMPLE_TABLE, RT_RNG_SCENE) of rt_rng.h:
    # choose_mat = draw 0, centre.x = -11 + 0.9 * draw 1, centre.z = -11 + 0.9 * draw 2
    first = a.desc.primitives[1]
    d1 = orc.lib().orc_rng_double(1, 1, 0xFFFFFFFE, 0, 7, 0, 0)
    d2 = orc.lib().orc_rng_double(1, 2, 0xFFFFFFFE, 0, 7, 0, 0)
    assert first.p[0] == -11.0 + 0.9 * d1 and first.p[2] == -11.0 + 0.9 * d2


def test_oracle_bvh_equals_linear_scan_on_random_scene(host, orc):
    s = random_session(host)
    p = s.params
    p.width, p.height, p.samples = 48, 27, 2
    cam = host.camera_new((0, 2, 10), (0, 0, 0), 20.0, 0.1, 10.0, p.width, p.height)
    a, sa = orc.render(s.desc, cam, p, use_bvh=1)
    b, sb = orc.render(s.desc, cam, p, use_bvh=0)
    assert np.array_equal(a, b) and sa == sb


# ------------------------------------------------------------------------- GPU
def many_boxes_scene(n=70, seed=3):
    """Rotated / translated boxes, rects and spheres of every material: BVH bounds of wrapped primitives."""
    rng = np.random.default_rng(seed)
    textures = [abi.solid(tuple(rng.random(3) * 0.8 + 0.1)) for _ in range(6)] + [abi.solid((6.0, 6.0, 6.0))]
    materials = [abi.material(S.L, i) for i in range(4)] + [abi.material(S.M, 4, fuzz=0.3), abi.material(S.M, 5, fuzz=0.0),
                                                            abi.material(S.D, -1, ior=1.5), abi.material(S.E, 6)]
    prims = [abi.rect(abi.RT_PRIM_XZ_RECT, -30, 30, -30, 30, 0.0, 0, 1)]
    for i in range(n):
        c = rng.random(3) * np.array([40.0, 6.0, 40.0]) - np.array([20.0, 0.0, 20.0])
        m = int(rng.integers(0, len(materials)))
        kind = i % 4
        if kind == 0:
            prims.append(abi.sphere(tuple(c + [0, 1, 0]), float(rng.random() + 0.3), m, i + 2))
        elif kind == 1:
            b = abi.box((0, 0, 0), tuple(rng.random(3) * 2 + 0.5), m, i + 2)
            ang = float(rng.random() * 2 * math.pi)
            b.flags = abi.RT_PRIM_HAS_ROTATE_Y | abi.RT_PRIM_HAS_TRANSLATE
            b.rot_sin, b.rot_cos = math.sin(ang), math.cos(ang)
            b.translate = abi.D3(*c)
            prims.append(b)
        elif kind == 2:
            r = abi.rect(abi.RT_PRIM_XY_RECT, c[0], c[0] + 2, c[1], c[1] + 2, c[2], m, i + 2)
            r.flags = abi.RT_PRIM_HAS_TRANSLATE
            r.translate = abi.D3(0.5, 0.25, -0.5)
            prims.append(r)
        else:
            prims.append(abi.moving_sphere(tuple(c + [0, 1, 0]), tuple(c + [0.5, 1.5, 0]), 0.6, m, i + 2))
    cam = dict(look_from=(0.0, 12.0, 45.0), look_at=(0.0, 2.0, 0.0), vfov=40.0, aperture=0.2, focus_distance=40.0)
    return abi.SceneBundle(prims, materials, textures, abi.sky()), cam


def same_frame(a, b):
    """Two closest-hit routines (separate template instances, different FMA contraction) trace the
    same paths up to rounding; where two surfaces touch (spheres resting on the ground) a hit can
    tip the other way for an isolated sample, as between device and oracle."""
    d = np.abs(a - b)
    return d.max() < 1e-3 and float((d.max(axis=-1) > 1e-9).mean()) < 2e-3


def render_gpu(rt, desc, cam, params, closest_hit=abi.RT_HIT_AUTO):
    """closest_hit: RtSceneOptions.closest_hit (rt_scene_create_ex)."""
    scene = rt.Scene(desc, closest_hit=closest_hit)
    try:
        return scene.render_frame(cam, params), scene.last_stats()
    finally:
        scene.close()


@pytest.mark.gpu
def test_gpu_random_scene_matches_oracle_and_linear_loop(rt, host, orc, gpu):
    s = random_session(host)
    p = s.params
    p.width, p.height, p.samples = 96, 54, 8
    cam = host.camera_new((0, 2, 10), (0, 0, 0), 20.0, 0.1, 10.0, p.width, p.height)
    ref, ref_segs = orc.render(s.desc, cam, p, use_bvh=1)
    bvh, st = render_gpu(rt, s, cam, p)                       # 485 primitives -> BVH variant
    lin, st_lin = render_gpu(rt, s, cam, p, abi.RT_HIT_LINEAR)  # same scene through the brute-force loop
    d = np.abs(s.tone_map(bvh) - s.tone_map(ref))
    assert d.max() < 1e-3 and float((d.max(axis=-1) > 1e-9).mean()) < 2e-3
    assert abs(int(st.segments) - ref_segs) <= 4
    # the two closest-hit routines are separate template instances (different FMA contraction),
    # so they agree to rounding, not bit for bit; the same paths are traced
    assert same_frame(bvh, lin) and abs(int(st.segments) - int(st_lin.segments)) <= 4


@pytest.mark.gpu
def test_gpu_bvh_with_wrapped_primitives(rt, orc, gpu):
    bundle, cam = many_boxes_scene()
    w, h, spp = 96, 54, 8
    camera = S.camera_for(cam, w, h)
    params = abi.render_params(w, h, spp)
    ref, ref_segs = orc.render(bundle.desc, camera, params, use_bvh=0)   # linear: the oracle's BVH mirrors RotateY's box bug
    bvh, st = render_gpu(rt, bundle, camera, params)
    lin, _ = render_gpu(rt, bundle, camera, params, abi.RT_HIT_LINEAR)
    d = np.abs(bvh - ref)
    assert d.max() < 1e-3 and float((d.max(axis=-1) > 1e-9).mean()) < 2e-3
    assert abs(int(st.segments) - ref_segs) <= 4
    assert same_frame(bvh, lin)


@pytest.mark.gpu
def test_gpu_small_scene_forced_through_bvh(rt, orc, gpu):
    """cornell_box and three_balls through the BVH variant: same frame as the linear loop."""
    for fn in (S.cornell_box, S.three_balls, S.cornell_box_boxes):
        bundle, cam, _ = fn()
        camera = S.camera_for(cam, 64, 36)
        params = abi.render_params(64, 36, 8)
        a, _ = render_gpu(rt, bundle, camera, params, abi.RT_HIT_LINEAR)
        b, _ = render_gpu(rt, bundle, camera, params, abi.RT_HIT_BVH)
        assert same_frame(a, b), fn.__name__


def moving_mix_scene(n=90, seed=11):
    """Static spheres, MovingSpheres on TWO different time intervals and a ground rect: the BVH walk tests leaves
    from its compact records (unwrapped spheres on the scene-wide interval — that of the first MovingSphere) or
    from the Prim records (the other interval, the rect)."""
    rng = np.random.default_rng(seed)
    textures = [abi.solid(tuple(rng.random(3) * 0.8 + 0.1)) for _ in range(5)]
    materials = [abi.material(S.L, i) for i in range(4)] + [abi.material(S.M, 4, fuzz=0.2)]
    prims = [abi.rect(abi.RT_PRIM_XZ_RECT, -30, 30, -30, 30, 0.0, 0, 1)]
    for i in range(n):
        c = rng.random(3) * np.array([24.0, 3.0, 24.0]) - np.array([12.0, -0.4, 12.0])
        m = int(rng.integers(0, len(materials)))
        r = float(rng.random() * 0.5 + 0.2)
        if i % 3 == 0:
            prims.append(abi.sphere(tuple(c), r, m, i + 2))
        else:
            interval = (0.0, 1.0) if i % 3 == 1 else (-0.5, 1.5)   # both contain the camera's shutter [0, 1]
            prims.append(abi.moving_sphere(tuple(c), tuple(c + rng.random(3) - 0.5), r, m, i + 2, *interval))
    bundle = abi.SceneBundle(prims, materials, textures, abi.sky())
    cam = dict(look_from=(0, 4, 18), look_at=(0, 1, 0), vfov=35.0, aperture=0.0, focus_distance=10.0)
    return bundle, cam


@pytest.mark.gpu
def test_gpu_bvh_leaf_records_with_two_time_intervals(rt, orc, gpu):
    bundle, cam = moving_mix_scene()
    w, h, spp = 96, 54, 8
    camera = S.camera_for(cam, w, h)
    params = abi.render_params(w, h, spp)
    ref, ref_segs = orc.render(bundle.desc, camera, params, use_bvh=0)
    bvh, st = render_gpu(rt, bundle, camera, params)
    lin, _ = render_gpu(rt, bundle, camera, params, abi.RT_HIT_LINEAR)
    assert same_frame(bvh, lin)
    d = np.abs(bvh - ref)
    assert d.max() < 1e-3 and float((d.max(axis=-1) > 1e-9).mean()) < 2e-3
    assert abs(int(st.segments) - ref_segs) <= 4


@pytest.mark.gpu
def test_gpu_bvh_with_axis_parallel_rays(rt, orc, gpu):
    """Direction components of EXACTLY zero through the f32 culling boxes (1/d = inf: the slab test once formed
    inf - inf = NaN there and culled the root, so such a ray silently missed the whole scene — rt_bvh_slab.h).
    An orthographic-style camera does it for every primary ray: `horizontal` only moves the ORIGIN-side corner
    along x while the direction's x stays 0 (upper_left_corner.x - origin.x == 0 and u * 0 == 0), and with
    vertical = 0 as well the second camera below makes every primary ray exactly (0, 0, -1)."""
    bundle, _ = moving_mix_scene(n=120, seed=4)
    w, h, spp = 96, 54, 4
    params = abi.render_params(w, h, spp, max_depth=6)
    for zero_axes in ("x", "xy"):
        cam = abi.RtCamera()
        cam.origin = abi.D3(0.25, 1.0, 18.0)
        cam.upper_left_corner = abi.D3(0.25, 4.0 if zero_axes == "x" else 1.0, 8.0)   # direction = ulc + u*h - v*vert - origin
        cam.horizontal = abi.D3(0.0, 0.0, 0.0)
        cam.vertical = abi.D3(0.0, 6.0 if zero_axes == "x" else 0.0, 0.0)
        cam.right, cam.up, cam.forward = abi.D3(1, 0, 0), abi.D3(0, 1, 0), abi.D3(0, 0, 1)
        cam.lens_radius, cam.focus_distance, cam.time_a, cam.time_b = 0.0, 10.0, 0.0, 1.0
        ref, ref_segs = orc.render(bundle.desc, cam, params, use_bvh=0)
        bvh, st = render_gpu(rt, bundle, cam, params, abi.RT_HIT_BVH)
        lin, _ = render_gpu(rt, bundle, cam, params, abi.RT_HIT_LINEAR)
        assert same_frame(bvh, lin), zero_axes
        d = np.abs(bvh - ref)
        assert d.max() < 1e-3 and float((d.max(axis=-1) > 1e-9).mean()) < 2e-3, zero_axes
        assert abs(int(st.segments) - ref_segs) <= 4
        assert int(st.segments) > w * h * spp * 1.2      # the rays do hit the scene and bounce (a culled root would give sky only)
