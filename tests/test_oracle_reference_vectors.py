"""Pins the CPU oracle to everything the reference itself holds for this path:

1. the reference's own unit tests (racer-tracer/src/vec3.rs:446-503) replayed
   on the oracle's Vec3 operators;
2. the published Philox4x32 known-answer vectors for 7 and 10 rounds (Random123 kat_vectors),
   since the RNG replaces the un-vendored, unseedable `rand 0.8.5`;
3. pixels of the reference's own output images (assets/*.png, committed as
   values in tests/golden/reference_assets.json): sky pixels must match
   exactly, block means statistically.

Everything else on the path is "parity unpinned" by reference tests (there are
none) and is covered by the hand-derived known answers in test_oracle_kats.py.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import scenes_py as S

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_assets.json")) as f:
    ASSETS = json.load(f)


def v3(orc, fn, a, b):
    out = S.abi.D3()
    getattr(orc.lib(), fn)(S.abi.D3(*a), S.abi.D3(*b) if not isinstance(b, float) else b, out)
    return list(out)


# ---- 1. vec3.rs:446-503 ----------------------------------------------------
def test_reference_vec3_add(orc):  # vec3.rs:449-460
    v1, v2 = (1.0, 2.0, 3.0), (2.0, 4.0, 6.0)
    assert v3(orc, "orc_vec3_add", v1, v2) == [3.0, 6.0, 9.0]
    assert v3(orc, "orc_vec3_add", v2, v1) == v3(orc, "orc_vec3_add", v1, v2)


def test_reference_vec3_sub(orc):  # vec3.rs:462-475
    v1, v2 = (1.0, 2.0, 3.0), (2.0, 4.0, 6.0)
    assert v3(orc, "orc_vec3_sub", v1, v2) == [-1.0, -2.0, -3.0]
    assert v3(orc, "orc_vec3_sub", v2, v1) == [1.0, 2.0, 3.0]


def test_reference_vec3_mul(orc):  # vec3.rs:477-492
    v1 = (1.0, -2.0, 3.0)
    assert v3(orc, "orc_vec3_scale", v1, 5.0) == [5.0, -10.0, 15.0]
    assert v3(orc, "orc_vec3_mul", v1, (4.0, 8.0, 16.0)) == [4.0, -16.0, 48.0]


def test_reference_vec3_div(orc):  # vec3.rs:494-502
    assert v3(orc, "orc_vec3_div", (1.0, -2.0, 3.0), 2.0) == [0.5, -1.0, 1.5]


# ---- 2. Philox4x32 known answers ---------------------------------------------
# Random123's kat_vectors lines "philox4x32 7 ..." and "philox4x32 10 ...": counter, key, output.
# The contract (include/rt_rng.h) runs 7 rounds; the 10-round vectors pin the same round function.
PHILOX_KATS = [
    (7, (0, 0, 0, 0), (0, 0), (0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48)),
    (7, (0xffffffff,) * 4, (0xffffffff,) * 2, (0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662)),
    (7, (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a)),
    (10, (0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    (10, (0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    (10, (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]
CONTRACT_ROUNDS = 7   # RT_PHILOX_ROUNDS


@pytest.mark.parametrize("rounds,ctr,key,expect", PHILOX_KATS)
def test_philox_known_answers(orc, rounds, ctr, key, expect):
    out = (C.c_uint32 * 4)()
    orc.lib().orc_philox4x32((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), rounds, out)
    assert tuple(out) == expect


def test_contract_rounds_match_the_header():
    import os
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rt_rng.h")).read()
    assert int(re.search(r"#define RT_PHILOX_ROUNDS (\d+)", text).group(1)) == CONTRACT_ROUNDS


def test_rng_double_is_53_bit_of_addressed_block(orc):
    """include/rt_rng.h: d0 = u53(out[0], out[1]), d1 = u53(out[2], out[3])."""
    seed, pixel, sample, seg, purpose, block = 0x0123456789ABCDEF, 77, 5, 3, 3, 9
    out = (C.c_uint32 * 4)()
    orc.lib().orc_philox4x32((C.c_uint32 * 4)(pixel, sample, (seg << 8) | purpose, block),
                             (C.c_uint32 * 2)(seed & 0xffffffff, seed >> 32), CONTRACT_ROUNDS, out)
    for which, (hi, lo) in enumerate(((out[0], out[1]), (out[2], out[3]))):
        want = float(((hi << 32 | lo) >> 11)) * 2.0 ** -53
        got = orc.lib().orc_rng_double(seed, pixel, sample, seg, purpose, block, which)
        assert got == want and 0.0 <= got < 1.0


def test_rng_triple_is_three_42_bit_draws_of_one_block(orc):
    """include/rt_rng.h: e_j = u42(out[j], out[3] >> 10 j) — the three coordinates of a random_in_unit_sphere candidate."""
    seed, pixel, sample, seg, purpose, block = 0x0123456789ABCDEF, 77, 5, 3, 3, 9
    out = (C.c_uint32 * 4)()
    orc.lib().orc_philox4x32((C.c_uint32 * 4)(pixel, sample, (seg << 8) | purpose, block),
                             (C.c_uint32 * 2)(seed & 0xffffffff, seed >> 32), CONTRACT_ROUNDS, out)
    e = (C.c_double * 3)()
    orc.lib().orc_rng_triple(seed, pixel, sample, seg, purpose, block, e)
    for j in range(3):
        want = float((out[j] << 10) | ((out[3] >> (10 * j)) & 0x3FF)) * 2.0 ** -42
        assert e[j] == want and 0.0 <= e[j] < 1.0


def test_rng_triple_candidates_fill_the_unit_sphere_like_independent_uniforms(orc):
    """The 42-bit coordinates share a block (and the ten low bits of each come from ONE word): accepted candidates
    must still be uniform in the ball — acceptance pi/6, zero mean, E[x^2] = 1/5, no correlation between axes."""
    e = (C.c_double * 3)()
    pts = np.empty((60000, 3))
    for i in range(len(pts)):
        orc.lib().orc_rng_triple(1, i, 3, 1, 3, 0, e)
        pts[i] = (-1.0 + 2.0 * e[0], -1.0 + 2.0 * e[1], -1.0 + 2.0 * e[2])
    assert abs(pts.mean(axis=0)).max() < 0.01 and abs(pts.var(axis=0) - 1 / 3).max() < 0.01
    inside = pts[(pts ** 2).sum(axis=1) < 1.0]
    assert abs(len(inside) / len(pts) - np.pi / 6) < 0.008                      # sigma = 0.002
    assert abs(inside.mean(axis=0)).max() < 0.01 and abs((inside ** 2).mean(axis=0) - 0.2).max() < 0.005
    assert abs(np.corrcoef(inside.T) - np.eye(3)).max() < 0.02
    # the ten low bits of the three coordinates are disjoint slices of out[3]: pairwise independent
    low = np.round((pts + 1.0) * 2.0 ** 41).astype(np.int64) & 0x3FF
    assert abs(np.corrcoef(low.T.astype(float)) - np.eye(3)).max() < 0.02


def test_rng_uniformity(orc):
    xs = np.array([orc.lib().orc_rng_double(1, i, 0, 0, 1, 0, 0) for i in range(20000)])
    assert abs(xs.mean() - 0.5) < 0.01 and abs(xs.var() - 1 / 12) < 0.005
    hist, _ = np.histogram(xs, bins=10, range=(0, 1))
    assert hist.min() > 1800 and hist.max() < 2200


def _philox_np(ctr, key, rounds):
    """Vectorised numpy Philox4x32 (a third implementation, only for the statistics below)."""
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    c = [np.asarray(x, dtype=np.uint64) & np.uint64(0xffffffff) for x in ctr]
    k0, k1 = int(key[0]), int(key[1])
    mask = np.uint64(0xffffffff)
    for _ in range(rounds):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & mask, (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & mask]
        k0, k1 = (k0 + W0) & 0xffffffff, (k1 + W1) & 0xffffffff
    return c


def test_numpy_philox_agrees_with_the_oracle(orc):
    out = (C.c_uint32 * 4)()
    orc.lib().orc_philox4x32((C.c_uint32 * 4)(123, 45, 678, 9), (C.c_uint32 * 2)(1, 0), CONTRACT_ROUNDS, out)
    got = _philox_np((np.array([123]), np.array([45]), np.array([678]), np.array([9])), (1, 0), CONTRACT_ROUNDS)
    assert [int(x[0]) for x in got] == list(out)


@pytest.mark.parametrize("word", [0, 1, 2, 3])
def test_contract_rounds_decorrelate_adjacent_addresses(word):
    """The draws of this path differ in ONE counter word by small steps (next pixel, next
    sample, next segment/purpose, next block).  With the contract's round count the outputs of
    such neighbours must look independent: flipping a low counter bit flips half of the 128
    output bits (avalanche), and the doubles of consecutive addresses are uncorrelated and
    uniform in 2-D."""
    n = 1 << 16
    base = [np.full(n, 1234567, dtype=np.uint64), np.full(n, 89, dtype=np.uint64),
            np.full(n, (3 << 8) | 3, dtype=np.uint64), np.full(n, 0, dtype=np.uint64)]
    ctr = [b.copy() for b in base]
    ctr[word] = ctr[word] + np.arange(n, dtype=np.uint64)
    out = _philox_np(ctr, (1, 0), CONTRACT_ROUNDS)
    # avalanche: neighbour (+1 in that word) differs in ~64 of 128 bits
    flipped = sum(np.unpackbits(((a[1:] ^ a[:-1]) & np.uint64(0xffffffff)).astype(">u4").view(np.uint8)).sum() for a in out)
    per_pair = flipped / (n - 1)
    assert abs(per_pair - 64.0) < 0.2                         # sigma of the mean ~ 0.022
    x = ((out[0] << np.uint64(32) | out[1]) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    y = ((out[2] << np.uint64(32) | out[3]) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    for u in (x, y):
        assert abs(u.mean() - 0.5) < 0.005 and abs(u.var() - 1.0 / 12.0) < 0.002
        assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 0.02   # lag-1 over consecutive addresses
    assert abs(np.corrcoef(x, y)[0, 1]) < 0.02
    grid, _, _ = np.histogram2d(x[:-1], x[1:], bins=16, range=((0, 1), (0, 1)))
    expect = (n - 1) / 256.0
    chi2 = ((grid - expect) ** 2 / expect).sum()
    assert chi2 < 255 + 6 * np.sqrt(2 * 255)                   # 255 dof, six sigma


# ---- 3. the reference's own output images -----------------------------------
def _pixel_mean(orc, bundle, cam, params, x, y, n):
    scene = orc.lib().orc_scene_build(C.byref(bundle.desc), 1, params.seed)
    acc = np.zeros(3)
    out = S.abi.D3()
    for s in range(n):
        orc.lib().orc_sample_radiance(C.byref(bundle.desc), scene, C.byref(cam), C.byref(params), x, y, s, out, None)
        acc += np.array(list(out))
    orc.lib().orc_scene_free(scene)
    return acc / n


def test_three_balls_sky_pixels_match_reference_png(orc):
    """assets/three_balls.png was written by the reference's SavePng at the
    default 600x600 (config.yml:24-26).  Its sky pixels pin camera basis,
    pixel->(u,v) incl. row 0 = top, Sky lerp, sqrt gamma, tone map None and
    the truncating *255."""
    bundle, cam, _ = S.three_balls()
    camera = S.camera_for(cam, 600, 600)
    params = S.abi.render_params(600, 600, 1)
    exact = 0
    for pt in ASSETS["sky_pixels"]["three_balls"]:
        mean = _pixel_mean(orc, bundle, camera, params, pt["x"], pt["y"], 32)
        val = np.sqrt(mean) * 255.0
        got = orc.pack_rgba8(np.sqrt(mean)[None, :])[0]
        frac = np.abs(val - np.round(val))
        want = np.array(pt["rgba"])
        assert got[3] == 255 == want[3]
        robust = (frac[:3] > 0.08) | (val[:3] >= 254.999)  # not sitting on a rounding edge
        assert (got[:3][robust] == want[:3][robust]).all(), (pt, val)
        # on an edge the jitter decides which side the reference landed on
        assert (np.abs(got[:3].astype(int) - want[:3]) <= 1).all(), (pt, val)
        exact += int(robust.sum())
    assert exact >= 60  # of 75 channel values


def test_noise_and_textures_sky_pixels_match_reference_png(orc):
    """Same check through the other scene's camera (noise_and_textures.yml:67-75);
    only its sky is usable, the Perlin sphere is randomly seeded in the reference."""
    textures = [S.abi.solid((0.5, 0.5, 0.5))]
    materials = [S.abi.material(S.L, 0)]
    prims = [S.abi.sphere((0.0, -1000.0, 0.0), 1000.0, 0, 1)]
    bundle = S.abi.SceneBundle(prims, materials, textures, S.abi.sky())
    camera = S.camera_for(dict(look_from=(12.3, 4.0, 9.7), look_at=(-0.3, 0.7, 0.0), vfov=20.0, aperture=0.01,
                               focus_distance=4.0), 600, 600)
    params = S.abi.render_params(600, 600, 1)
    for pt in ASSETS["sky_pixels"]["noise_and_textures"]:
        mean = _pixel_mean(orc, bundle, camera, params, pt["x"], pt["y"], 16)
        got = orc.pack_rgba8(np.sqrt(mean)[None, :])[0]
        assert (np.abs(got[:3].astype(int) - np.array(pt["rgba"][:3])) <= 1).all(), pt


ROOT = os.path.dirname(HERE)
ASSET_SPP = 64   # the screenshots were saved at 200 spp; a 75x75-pixel block of a 64-spp render is already converged to ~1e-3


def _render_like_the_reference(orc, scene):
    """The oracle on a shipped scene file under the reference's own default config (config.yml: 600x600,
    depth 20), through the C++ loader's PODs -> (float frame, quantised RGB in [0,1] like the PNG, session)."""
    import importlib
    host = importlib.import_module("racer-tracer_amd.host")
    s = host.Session(os.path.join(ROOT, "scenes", "config_ref.yml"), scene=os.path.join(ROOT, "scenes", scene + ".yml"))
    p = s.params
    assert (p.width, p.height, p.max_depth) == (600, 600, 20)
    p.samples = ASSET_SPP
    frame, _ = orc.render(s.desc, s.camera, p)
    otm = orc.OrcToneMap.from_buffer_copy(s.tone_map_desc)
    mapped = np.empty_like(frame)
    orc.lib().orc_tone_map_apply(C.byref(otm), frame.ctypes.data_as(C.POINTER(C.c_double)),
                                 mapped.ctypes.data_as(C.POINTER(C.c_double)), frame.size // 3)
    return frame, orc.pack_rgba8(mapped)[..., :3].astype(np.float64) / 255.0, s


def _blocks(q, n):
    b = 600 // n
    return q.reshape(n, b, n, b, 3).mean(axis=(1, 3))


def test_three_balls_block_means_match_reference_png(orc):
    """assets/three_balls.png vs the oracle at 4x4 and 8x8 blocks.  three_balls.yml is fully deterministic
    (no Perlin table), so the only differences are Monte Carlo noise and the reference's other random
    stream: Lambertian, Dielectric (incl. the negative-radius shell), fuzz-0 Metal, Sky, the thin lens."""
    _, q, s = _render_like_the_reference(orc, "three_balls")
    assert s.tone_map_name == "None"
    d4 = np.abs(_blocks(q, 4) - np.array(ASSETS["block_means"]["three_balls"]))
    d8 = np.abs(_blocks(q, 8) - np.array(ASSETS["block_means_8x8"]["three_balls"]))
    assert d4.max() < 0.004, d4.max()       # measured 0.0011
    assert d8.max() < 0.008, d8.max()       # measured 0.0036
    assert d4[[0, 3]].max() < 0.001         # pure sky / pure ground rows


def test_clown_block_means_match_reference_png(orc):
    """assets/clown.png: 23 spheres (a real BVH in the reference), Lambertian + Dielectric + Metal with
    fuzz 0.2 (clown.yml:39,44 - the one shipped use of a fuzzy mirror, metal.rs:26-43), tone map None.
    Fully deterministic scene: the oracle reproduces the screenshot's block means to a few 1e-4."""
    _, q, s = _render_like_the_reference(orc, "clown")
    assert s.tone_map_name == "None" and s.desc.n_primitives == 23
    d4 = np.abs(_blocks(q, 4) - np.array(ASSETS["block_means"]["clown"]))
    d8 = np.abs(_blocks(q, 8) - np.array(ASSETS["block_means_8x8"]["clown"]))
    assert d4.max() < 0.002, d4.max()       # measured 0.0004
    assert d8.max() < 0.004, d8.max()       # measured 0.0009


def test_noise_and_textures_blocks_outside_the_marble_sphere_match_reference_png(orc):
    """assets/noise_and_textures.png: the Perlin sphere's gradient table is drawn from thread_rng in the
    reference (noise.rs:45-47), so the blocks it covers cannot match; every other 75x75 block must -
    the Checkered ground (checkered.rs:32-42), the TextureImage earth incl. its JPEG decode
    (texture/image.rs:28-51), the glass sphere, the sky.  Blocks are chosen by projecting the marble
    sphere (centre (0,1,0), r = 1) through the scene's camera."""
    _, q, s = _render_like_the_reference(orc, "noise_and_textures")
    cam = s.camera
    d = np.array([0.0, 1.0, 0.0]) - np.array(list(cam.origin))
    x, y, z = d @ np.array(list(cam.right)), d @ np.array(list(cam.up)), -(d @ np.array(list(cam.forward)))
    cx, cy = (0.5 + x / z / cam.viewport_width) * 600, (0.5 - y / z / cam.viewport_height) * 600
    radius = np.tan(np.arcsin(1.0 / np.linalg.norm(d))) / cam.viewport_height * 600 + 12     # + defocus and pixel filter
    assert abs(cx - 319.5) < 1 and abs(cy - 273.8) < 1 and abs(radius - 118.9) < 1
    d8 = np.abs(_blocks(q, 8) - np.array(ASSETS["block_means_8x8"]["noise_and_textures"])).max(axis=-1)
    checked = 0
    for by in range(8):
        for bx in range(8):
            nx, ny = np.clip(cx, bx * 75, bx * 75 + 75), np.clip(cy, by * 75, by * 75 + 75)   # nearest point of the block
            if (nx - cx) ** 2 + (ny - cy) ** 2 <= radius ** 2:
                continue
            checked += 1
            assert d8[by, bx] < 0.012, (by, bx, d8[by, bx])     # measured <= 0.009 (a glass block that refracts the marble)
    assert checked >= 48
    assert d8[[0, 1, 6, 7]].max() < 0.003                       # sky rows and the far checker ground
    d4 = np.abs(_blocks(q, 4) - np.array(ASSETS["block_means"]["noise_and_textures"])).max(axis=-1)
    assert d4[[0, 3]].max() < 0.003
    # the earth sphere (projected centre (475, 251), r = 99 px) fills columns 6-7 of rows 2-4: checked above, and not flat
    assert d8[2:5, 6:8].max() < 0.012 and q[225:300, 450:525].std() > 0.02


def test_marble_statistics_of_the_reference_png_lie_inside_the_seed_envelope(orc):
    """The VALUE side of Noise (noise.rs:26-33: color * 0.5 * (1 + sin(scale * z + 10 * turb))), which no fixed table
    can match pixel for pixel: eight Perlin seeds through the oracle, the screenshot's marble statistics inside their
    envelope — and, so that the check demonstrably has teeth, renders WITHOUT the sine's z phase and with a wrong
    scale outside it (tests/noise_stats.py says what this pins and what it cannot)."""
    import importlib
    import noise_stats as N
    host = importlib.import_module("racer-tracer_amd.host")

    def stats(session):
        p = session.params
        assert (p.width, p.height) == (600, 600) and session.tone_map_name == "None"
        p.samples = 16
        frame, _ = orc.render(session.desc, session.camera, p)
        return N.patch_stats(orc.pack_rgba8(frame)[..., :3].astype(np.float64) / 255.0)

    seeds = np.array([stats(N.noise_session(host, seed)) for seed in range(1, 9)])
    ref = N.reference_stats()
    ok = N.inside_envelope(ref, seeds)
    assert ok.all(), (ref[~ok], seeds.min(axis=0)[~ok], seeds.max(axis=0)[~ok])
    # teeth: the same renderer with the formula broken must fall OUTSIDE the envelope
    no_phase = stats(N.noise_session(host, 1, scale=0.0))      # sin(10 * turb) only
    assert not N.inside_envelope(no_phase, seeds)[:3].any()    # the mean colour is far off (0.63 against 0.39 .. 0.50)
    wrong_scale = stats(N.noise_session(host, 2, scale=12.0))
    assert not N.inside_envelope(wrong_scale, seeds)[-3:].any()   # contrast moves to the small cells (std5 / std35 2.1 against 1.2 .. 1.55)


def test_emissive_lights_and_background_match_reference_png(orc):
    """assets/emissive.png cannot be matched in VALUE: its lights saturate to (255,255,254) where the shipped
    emissive.yml's (1,1,1) and (4,4,4) emitters give Aces(1) -> 157 and Aces(2) -> 204 (the screenshot
    predates the scene file, like cornell_box.png), and every diffuse surface is a randomly seeded Noise.
    What it does pin is WHERE things are: the sphere light and the rect light (DiffuseLight on a Sphere and
    an XyRect, diffuse_light.rs:25-37, seen through the thin-lens camera) cover the same pixels, row by row
    and column by column, and the background is black in the same places."""
    frame, q, s = _render_like_the_reference(orc, "emissive")
    assert s.tone_map_name == "Aces"
    lay = ASSETS["emissive_layout"]
    q8 = np.round(q * 255.0).astype(int)
    sphere_light = orc.pack_rgba8(orc.tone_map(orc.ORC_TM_ACES, np.array([[1.0, 1.0, 1.0]])))[0, :3]   # sqrt(1) tone-mapped
    rect_light = orc.pack_rgba8(orc.tone_map(orc.ORC_TM_ACES, np.array([[2.0, 2.0, 2.0]])))[0, :3]     # sqrt(4)
    assert list(sphere_light) == [157, 157, 157] and list(rect_light) == [204, 204, 204]
    lit = (q8 == sphere_light).all(axis=-1) | (q8 == rect_light).all(axis=-1)   # pixels whose every sample saw a light directly
    assert abs(int(lit.sum()) - lay["lit_pixels"]) < 0.01 * lay["lit_pixels"]     # 19784 vs 19717
    rows = lit.sum(axis=1) - np.array(lay["lit_row_counts"])
    cols = lit.sum(axis=0) - np.array(lay["lit_col_counts"])
    assert np.abs(rows).max() <= 4                                             # measured 3
    # the rect light's vertical edges fall between pixel columns: a whole column of ~100 pixels flips with the jitter
    assert np.abs(cols).max() <= 16 and (np.abs(cols) > 4).sum() <= 8           # measured 13, 5 columns
    assert (rows != 0).sum() + (cols != 0).sum() < 300
    bg = np.array(lay["black_grid"])
    assert len(bg) >= 100
    off = (frame[bg[:, 1], bg[:, 0]] != 0).any(axis=-1)
    assert off.sum() <= 2                                                      # measured 1: a stray bounce in 64 spp


def test_cornell_layout_matches_reference_png(orc):
    """assets/cornell_box.png pins geometry and hue only (its light intensity
    predates the v1 scene file): black outside columns
    15-585, green wall left, red wall right."""
    bundle, cam, _ = S.cornell_box()
    camera = S.camera_for(cam, 600, 600)
    frame, _ = orc.render(bundle.desc, camera, S.abi.render_params(600, 600, 8))
    tm = orc.tone_map(orc.ORC_TM_ACES, frame)
    assert (frame[:, :13] == 0).all() and (frame[:, 588:] == 0).all()
    assert frame[250:350, 16:30].sum() > 0 and frame[250:350, 570:584].sum() > 0
    mine = np.clip(tm, 0, 1).reshape(4, 150, 4, 150, 3).mean(axis=(1, 3))
    ref = np.array(ASSETS["block_means"]["cornell_box"])
    for m in (mine, ref):
        assert (m[:, 0, 1] > m[:, 0, 0]).all() and (m[:, 0, 1] > m[:, 0, 2]).all()   # left column: green dominates
        assert (m[:, 3, 0] > m[:, 3, 1]).all() and (m[:, 3, 0] > m[:, 3, 2]).all()   # right column: red dominates
    # the light rectangle saturates: Aces(sqrt(15)) * 255 truncates to 230
    # (the light rect, y = 554, z in [227, 332], projects to rows ~80-98 around column 300)
    assert list(orc.pack_rgba8(tm[np.newaxis, 90:91, 300])[0, 0]) == [230, 230, 230, 255]
