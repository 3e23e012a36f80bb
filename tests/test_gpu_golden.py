"""GPU vs the COMMITTED golden frames (tests/golden/oracle_frames.npz) for all
seven shipped scenes, driven through the C++ scene loader and the C ABI — the
same route the CLI and bench.py take.  Also the post path on real renders."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_oracle_fixtures as fx  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "oracle_frames.npz"))
TOL = 1e-3      # north star: per-channel |delta| < 1e-3 on the tone-mapped image
TIGHT = 1e-9


@pytest.fixture(scope="module")
def host():
    return importlib.import_module("racer-tracer_amd.host")


@pytest.mark.parametrize("name", sorted(fx.SCENES))
def test_gpu_matches_committed_oracle_frame(rt, host, gpu, name):
    s, p, _ = fx.load(host, name)
    cam = fx.camera_for(host, s, p)
    scene = rt.Scene(s)
    try:
        got = scene.render_frame(cam, p)
        stats = scene.last_stats()
    finally:
        scene.close()
    gold = GOLD[name + "_frame"]
    d = np.abs(s.tone_map(got) - s.tone_map(gold))
    d = np.where(np.isnan(d), 0.0, d)   # Reinhard-style 0/0 on black pixels is NaN on both sides
    assert np.isfinite(got).all()
    assert d.max() < TOL, (name, d.max())
    assert float((d.max(axis=-1) > TIGHT).mean()) < 2e-3, name
    assert abs(int(stats.segments) - int(GOLD[name + "_segments"])) <= 4


def test_noise_and_textures_full_resolution_band(rt, host, orc, gpu):
    """BASELINE config 4 at full width, reduced spp: Perlin + image texture +
    checker + glass against the live oracle on a band of rows."""
    s = host.Session(os.path.join(ROOT, "scenes", "config_c4.yml"),
                     scene=os.path.join(ROOT, "scenes", "noise_and_textures.yml"))
    p = s.params
    p.samples = 4
    scene = rt.Scene(s)
    try:
        got = scene.render_frame(s.camera, p)
    finally:
        scene.close()
    p.strip_rows, p.strip_count, p.strip_index = 8, 45, 22     # every 45th strip: 3 bands of 8 rows
    ref, _ = orc.render(s.desc, s.camera, p)
    rows = ((np.arange(p.height) // 8) % 45) == 22
    d = np.abs(ref[rows] - got[rows])
    assert d.max() < TOL
    assert float((d.max(axis=-1) > TIGHT).mean()) < 2e-3


def test_config3_at_full_size_and_full_spp_on_a_band(rt, host, orc, gpu):
    """BASELINE config 3 exactly as benchmarked (cornell_box 1920x1080, 1024 spp, depth 20): the
    device renders the whole frame, the oracle one 8-row band through the box;
    32 sample chunks summed per pixel against the oracle's one running sum."""
    s = host.Session(os.path.join(ROOT, "scenes", "config_c3.yml"), scene=os.path.join(ROOT, "scenes", "cornell_box.yml"))
    p = s.params
    assert (p.width, p.height, p.samples, p.max_depth) == (1920, 1080, 1024, 20)
    scene = rt.Scene(s)
    try:
        got = scene.render_frame(s.camera, p)
        stats = scene.last_stats()
    finally:
        scene.close()
    assert stats.samples == 1920 * 1080 * 1024
    p.strip_rows, p.strip_count, p.strip_index = 8, 135, 40          # rows 320..327: back wall, side walls, floor bounce
    ref, _ = orc.render(s.desc, s.camera, p)
    rows = ((np.arange(p.height) // 8) % 135) == 40
    d = np.abs(ref[rows] - got[rows])
    assert d.max() < TOL
    assert float((d.max(axis=-1) > TIGHT).mean()) < 2e-3
    assert got[rows].max() > 0.4 and (got[rows] > 0).mean() > 0.4      # the band crosses the lit box


def test_three_balls_full_resolution_band(rt, host, orc, gpu):
    """BASELINE config 2 at full size (1920x1080, aperture 0.1: the lens disk through the batched
    cooperative sampler; glass incl. the negative-radius sphere; metal), reduced spp, against
    the live oracle on bands of rows through the spheres."""
    s = host.Session(os.path.join(ROOT, "scenes", "config_c2.yml"),
                     scene=os.path.join(ROOT, "scenes", "three_balls.yml"))
    p = s.params
    assert (p.width, p.height) == (1920, 1080)
    p.samples = 6
    scene = rt.Scene(s)
    try:
        got = scene.render_frame(s.camera, p)
        stats = scene.last_stats()
    finally:
        scene.close()
    p.strip_rows, p.strip_count, p.strip_index = 8, 27, 17     # every 27th strip: 5 bands of 8 rows
    ref, ref_segs = orc.render(s.desc, s.camera, p)
    rows = ((np.arange(p.height) // 8) % 27) == 17
    d = np.abs(ref[rows] - got[rows])
    assert d.max() < TOL
    assert float((d.max(axis=-1) > TIGHT).mean()) < 2e-3
    assert 1.5 < stats.segments / stats.samples < 2.5


def test_render_to_png_end_to_end(rt, host, gpu, tmp_path):
    """renderer -> tone map -> SavePng, as main.rs:148-158 wires it."""
    from PIL import Image
    s, p, _ = fx.load(host, "cornell_box")
    cam = fx.camera_for(host, s, p)
    scene = rt.Scene(s)
    try:
        tiles = scene.render_tiles(cam, p)
    finally:
        scene.close()
    frame = np.zeros((p.height, p.width, 3))
    for r, c, w, h, arr in tiles:                      # ScreenBuffer::update (image_buffer.rs:135-170)
        frame[r:r + h, c:c + w] = s.tone_map(arr)
    path = s.save_png(frame, str(tmp_path))
    img = np.array(Image.open(path))
    gold = host.pack_rgba8(s.tone_map(GOLD["cornell_box_frame"]))
    assert img.shape == gold.shape
    assert np.abs(img.astype(int) - gold.astype(int)).max() <= 1   # truncation may flip an LSB at 1e-13 differences
    assert (img != gold).mean() < 1e-3


# ---- the HIP path against the reference's OWN output, no oracle in between -------------------------------
# assets/*.png are SavePng screenshots of the reference at its default config (600x600, 200 spp, depth 20);
# their block means are committed in tests/golden/reference_assets.json.  The oracle is pinned to them in
# tests/test_oracle_reference_vectors.py; here the device renders the same scenes at the same settings.
import json

with open(os.path.join(ROOT, "tests", "golden", "reference_assets.json")) as f:
    ASSETS = json.load(f)


def _gpu_screenshot(rt, host, scene):
    s = host.Session(os.path.join(ROOT, "scenes", "config_ref.yml"), scene=os.path.join(ROOT, "scenes", scene + ".yml"))
    p = s.params
    assert (p.width, p.height, p.samples, p.max_depth) == (600, 600, 200, 20)
    dev = rt.Scene(s)
    try:
        rgba = dev.render_frame_rgba8(s.camera, p, s.tone_map_desc)   # render + tone map + pack on the device
    finally:
        dev.close()
    return rgba[..., :3].astype(np.float64) / 255.0, s


def _blocks(q, n):
    b = 600 // n
    return q.reshape(n, b, n, b, 3).mean(axis=(1, 3))


@pytest.mark.parametrize("scene,tol4,tol8", [("clown", 0.002, 0.004), ("three_balls", 0.003, 0.006)])
def test_gpu_reproduces_the_reference_screenshot(rt, host, gpu, scene, tol4, tol8):
    """Deterministic scenes: the device's 600x600x200spp frame, tone-mapped and packed on the device, has the
    screenshot's block means (clown: fuzzy Metal + glass + 23 spheres; three_balls: thin lens, hollow glass)."""
    q, _ = _gpu_screenshot(rt, host, scene)
    d4 = np.abs(_blocks(q, 4) - np.array(ASSETS["block_means"][scene]))
    d8 = np.abs(_blocks(q, 8) - np.array(ASSETS["block_means_8x8"][scene]))
    assert d4.max() < tol4 and d8.max() < tol8, (d4.max(), d8.max())


def test_gpu_reproduces_noise_and_textures_outside_the_marble_sphere(rt, host, gpu):
    """Checker ground, earth image, glass and sky of assets/noise_and_textures.png; the blocks the randomly
    seeded Perlin sphere covers (projected centre (319.5, 273.8), radius 119 px incl. defocus) are left out."""
    q, _ = _gpu_screenshot(rt, host, "noise_and_textures")
    d8 = np.abs(_blocks(q, 8) - np.array(ASSETS["block_means_8x8"]["noise_and_textures"])).max(axis=-1)
    cx, cy, radius = 319.5, 273.8, 118.9
    checked = 0
    for by in range(8):
        for bx in range(8):
            nx, ny = np.clip(cx, bx * 75, bx * 75 + 75), np.clip(cy, by * 75, by * 75 + 75)
            if (nx - cx) ** 2 + (ny - cy) ** 2 <= radius ** 2:
                continue
            checked += 1
            assert d8[by, bx] < 0.012, (by, bx, d8[by, bx])
    assert checked >= 48 and d8[[0, 1, 6, 7]].max() < 0.003


def test_gpu_marble_statistics_of_the_reference_png_lie_inside_the_seed_envelope(rt, host, gpu):
    """The value side of the Noise texture on the DEVICE (wave-cooperative turbulence, lean sin, LDS gradient table):
    eight Perlin seeds rendered, tone-mapped and packed on the GPU; the screenshot's marble statistics must lie
    inside their envelope, and a render without the sine's z phase / with a wrong scale outside it
    (tests/noise_stats.py: what this pins, and that octaves 3-7 stay below what the screenshot shows)."""
    import noise_stats as N

    def stats(session):
        p = session.params
        p.samples = 64
        dev = rt.Scene(session)
        try:
            rgba = dev.render_frame_rgba8(session.camera, p, session.tone_map_desc)
        finally:
            dev.close()
        return N.patch_stats(rgba[..., :3].astype(np.float64) / 255.0)

    seeds = np.array([stats(N.noise_session(host, seed)) for seed in range(1, 9)])
    ref = N.reference_stats()
    ok = N.inside_envelope(ref, seeds)
    assert ok.all(), (ref[~ok], seeds.min(axis=0)[~ok], seeds.max(axis=0)[~ok])
    assert not N.inside_envelope(stats(N.noise_session(host, 1, scale=0.0)), seeds)[:3].any()
    assert not N.inside_envelope(stats(N.noise_session(host, 2, scale=12.0)), seeds)[-3:].any()
