"""Host layer (C++ inside the shipped library): YAML subset, Config/Args
precedence, scene loader -> POD tables, camera merge.  CPU only."""
import importlib
import math
import os
import subprocess
import textwrap

import numpy as np
import pytest

import scenes_py as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")
abi = S.abi


@pytest.fixture(scope="module")
def host():
    return importlib.import_module("racer-tracer_amd.host")


def open_scene(host, scene, config="config_c3.yml", **kw):
    return host.Session(os.path.join(SCENES, config), scene=os.path.join(SCENES, scene), **kw)


def prim_tuple(p):
    return (p.kind, p.flags, tuple(p.p), p.rot_sin, p.rot_cos, tuple(p.translate))


def material_key(desc, mi):
    m = desc.materials[mi]
    tex = None
    if m.kind != abi.RT_MAT_DIELECTRIC:
        t = desc.textures[m.texture]
        tex = (t.kind, tuple(t.color))
    return (m.kind, m.fuzz, m.refraction_index, tex)


@pytest.mark.parametrize("name", ["three_balls", "two_balls", "cornell_box", "cornell_box_boxes"])
def test_loader_matches_hand_transcription(host, name):
    """The loader's POD output == the independently hand-written tables of
    tests/scenes_py.py (geometry in file order, materials by value)."""
    s = open_scene(host, name + ".yml")
    bundle, cam, tm = getattr(S, name)()
    d, e = s.desc, bundle.desc
    assert d.n_primitives == e.n_primitives
    for i in range(d.n_primitives):
        got, want = prim_tuple(d.primitives[i]), prim_tuple(e.primitives[i])
        assert got[:3] == want[:3], (i, got, want)
        assert got[3:5] == pytest.approx(want[3:5], abs=1e-15) and got[5] == want[5]
        assert material_key(d, d.primitives[i].material) == material_key(e, e.primitives[i].material)
    assert d.background.kind == e.background.kind
    assert list(d.background.top) == list(e.background.top)
    assert list(d.background.bottom) == list(e.background.bottom) or d.background.kind == abi.RT_BG_SOLID
    assert s.tone_map_name == tm
    # camera: scene values over config values (camera.rs:403-435), 16:9 from config_c3
    want_cam = S.camera_for(cam, 1920, 1080)
    for f in ("origin", "upper_left_corner", "forward", "right", "up", "horizontal", "vertical"):
        assert list(getattr(s.camera, f)) == list(getattr(want_cam, f)), f
    assert s.camera.lens_radius == want_cam.lens_radius and s.camera.vfov == want_cam.vfov


def test_shared_materials_and_textures_are_emitted_once(host):
    s = open_scene(host, "cornell_box.yml")
    assert (s.desc.n_materials, s.desc.n_textures) == (4, 4)      # `white` is used by three rects
    s = open_scene(host, "clown.yml")
    assert s.desc.n_primitives == 23 and s.desc.n_materials == 5


def test_noise_and_textures_tables(host):
    s = open_scene(host, "noise_and_textures.yml", "config_c4.yml")
    d = s.desc
    assert (d.n_primitives, d.n_images, d.n_perlins) == (4, 1, 1)
    kinds = sorted(d.textures[i].kind for i in range(d.n_textures))
    assert kinds == [abi.RT_TEX_SOLID_COLOR, abi.RT_TEX_SOLID_COLOR, abi.RT_TEX_CHECKERED, abi.RT_TEX_IMAGE, abi.RT_TEX_NOISE]
    chk = [d.textures[i] for i in range(d.n_textures) if d.textures[i].kind == abi.RT_TEX_CHECKERED][0]
    assert list(d.textures[chk.tex_even].color) == [0.5, 1.0, 0.5]      # texture_a
    assert list(d.textures[chk.tex_odd].color) == [0.8, 0.8, 0.8]       # texture_b
    noise = [d.textures[i] for i in range(d.n_textures) if d.textures[i].kind == abi.RT_TEX_NOISE][0]
    assert (noise.scale, noise.depth, list(noise.color)) == (4.0, 7, [1.0, 1.0, 1.0])
    assert (d.images[0].width, d.images[0].height) == (1024, 512)
    assert s.tone_map_name == "None" and s.params.samples == 512
    # Perlin: 256 unit gradients, identity permutations (noise.rs:121-130)
    pl = d.perlins[0]
    g = np.array([[pl.ranvec[i][k] for k in range(3)] for i in range(256)])
    assert np.allclose(np.linalg.norm(g, axis=1), 1.0, atol=1e-14)
    assert list(pl.perm_x) == list(range(256)) == list(pl.perm_y) == list(pl.perm_z)
    assert np.abs(g.mean(axis=0)).max() < 0.15 and len(np.unique(np.round(g, 12), axis=0)) == 256


def test_perlin_table_follows_the_rng_contract(host, orc):
    """host/scene.cpp's own Philox must address draws as include/rt_rng.h says."""
    s = open_scene(host, "noise_and_textures.yml", "config_c4.yml", seed=77)
    pl = s.desc.perlins[0]
    for i in (0, 1, 100, 255):
        v = [-1.0 + 2.0 * orc.lib().orc_rng_double(77, i, 0xFFFFFFFE, 0, 5, blk, w)
             for blk, w in ((0, 0), (0, 1), (1, 0))]
        n = math.sqrt(sum(c * c for c in v))
        assert [pl.ranvec[i][k] for k in range(3)] == [c / n for c in v]
    other = open_scene(host, "noise_and_textures.yml", "config_c4.yml", seed=78)
    assert other.desc.perlins[0].ranvec[0][0] != pl.ranvec[0][0]


def test_emissive_scene_uses_its_own_tone_map_and_background(host):
    s = open_scene(host, "emissive.yml")
    assert s.tone_map_name == "Aces" and s.desc.background.kind == abi.RT_BG_SOLID
    kinds = sorted(s.desc.materials[i].kind for i in range(s.desc.n_materials))
    assert kinds == [abi.RT_MAT_LAMBERTIAN, abi.RT_MAT_DIFFUSE_LIGHT, abi.RT_MAT_DIFFUSE_LIGHT]


def test_render_params_come_from_the_config_file(host):
    s = open_scene(host, "three_balls.yml", "config_c1.yml")
    p = s.params
    assert (p.width, p.height, p.samples, p.max_depth, p.tiles_w, p.tiles_h, p.seed) == (400, 225, 16, 20, 10, 10, 1)
    q = s.preview_params
    assert (q.samples, q.max_depth) == (40, 10)
    assert s.image_action == 0 and s.image_output_dir == "./"


def write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(textwrap.dedent(text))
    return str(p)


MINI_CONFIG = """
    render:
      samples: 4
      max_depth: 5
      scale: 1
      num_threads_width: 2
      num_threads_height: 3
    screen:
      width: 40
      height: 20
    loader:
      Yml:
        path: %s
    image_action: SavePng
    tone_map:
      Reinhard:
        max_white: 4
    camera:
      vfov: 33
      focus_distance: 7
"""

MINI_SCENE = """
    textures:
      t:
        SolidColor:
          color:
            color: [0.5, 0.25, 1]   # comment after a value
    materials:
      m:
        Lambertian:
          texture_key: t
    geometry:
      ball:
        Sphere:
          pos: [1, 2, 3]
          radius: 2
          material: m
    camera:
      vfov: 50
      pos:
        pos: [0, 0, 9]
"""


def test_config_defaults_and_precedence(host, tmp_path):
    scene = write(tmp_path, "s.yml", MINI_SCENE)
    cfg = write(tmp_path, "c.yml", MINI_CONFIG % scene)
    s = host.Session(cfg)                                   # loader from the config file
    assert s.desc.n_primitives == 1 and s.image_action == 1 and s.tone_map_name == "Reinhard"
    assert (s.params.tiles_w, s.params.tiles_h, s.params.samples) == (2, 3, 4)
    # camera: scene vfov (50) beats config (33); config focus (7) fills the gap;
    # look_at falls to the default (0,0,-1); aperture default 0 (camera.rs:437-455)
    assert s.camera.vfov == 50.0 and s.camera.focus_distance == 7.0 and s.camera.lens_radius == 0.0
    assert list(s.camera.origin) == [0.0, 0.0, 9.0] and list(s.camera.forward) == [0.0, 0.0, 1.0]
    # CLI --image-action beats the file (config.rs:33-36); unknown strings mean None (config.rs:122-127)
    assert host.Session(cfg, image_action="none").image_action == 0
    assert host.Session(cfg, image_action="show").image_action == 0
    assert host.Session(cfg, image_action="png").image_action == 1
    # `texture_key` is accepted as well as its alias `texture` (yml.rs:69-83)
    assert list(s.desc.textures[0].color) == [0.5, 0.25, 1.0]
    tm = s.tone_map(np.array([[1.0, 1.0, 1.0]]))
    assert tm[0] == pytest.approx([(1 + 1 / 16) / 2] * 3)   # Reinhard, max_white 4


@pytest.mark.parametrize("mutation,code", [
    (("material: m", "material: nope"), 4),                                 # UnknownMaterial
    (("texture_key: t", "texture_key: nope"), 9),                           # SceneLoad: missing texture
    (("radius: 2", "radius: two"), 3),                                       # Configuration: not a number
    (("Sphere:", "Torus:"), 3),                                              # unknown variant
    (("geometry:", "geometri:"), 3),                                         # missing section
])
def test_loader_error_codes(host, tmp_path, mutation, code):
    scene = write(tmp_path, "s.yml", MINI_SCENE.replace(*mutation))
    cfg = write(tmp_path, "c.yml", MINI_CONFIG % scene)
    with pytest.raises(host.HostError) as e:
        host.Session(cfg)
    assert e.value.code == code, str(e.value)


def test_transform_wrappers_need_their_child(host, tmp_path):
    spin = "    geometry:\n      spin:\n        RotateY:\n          key: %s\n          degrees: 90\n"
    scene = write(tmp_path, "s.yml", MINI_SCENE.replace("    geometry:\n", spin % "missing"))
    cfg = write(tmp_path, "c.yml", MINI_CONFIG % scene)
    with pytest.raises(host.HostError) as e:
        host.Session(cfg)
    assert e.value.code == 9 and "did not have any child" in str(e.value)
    # with the right key the wrapper replaces its child under the child's name (yml.rs:401-419)
    scene = write(tmp_path, "s2.yml", MINI_SCENE.replace("    geometry:\n", spin % "ball"))
    s = host.Session(write(tmp_path, "c2.yml", MINI_CONFIG % scene))
    p = s.desc.primitives[0]
    assert s.desc.n_primitives == 1 and p.flags == abi.RT_PRIM_HAS_ROTATE_Y and p.kind == abi.RT_PRIM_SPHERE
    assert p.rot_sin == pytest.approx(1.0) and p.rot_cos == pytest.approx(0.0, abs=1e-16)


def test_missing_image_and_missing_files(host, tmp_path):
    scene = write(tmp_path, "s.yml", MINI_SCENE.replace("SolidColor:\n          color:\n            color: [0.5, 0.25, 1]   # comment after a value",
                                                         "Image:\n          path: /no/such/file.jpg"))
    cfg = write(tmp_path, "c.yml", MINI_CONFIG % scene)
    with pytest.raises(host.HostError) as e:
        host.Session(cfg)
    assert e.value.code == 21                                                # FailedToOpenImage
    with pytest.raises(host.HostError) as e:
        host.Session(str(tmp_path / "absent.yml"))
    assert e.value.code == 3                                                 # Configuration
    with pytest.raises(host.HostError) as e:
        host.Session(cfg, scene="scene.txt")
    assert e.value.code == 10                                                # ArgumentParsingError (config.rs:54-59)
    with pytest.raises(host.HostError) as e:
        host.Session(cfg, scene="noextension")
    assert e.value.code == 10
    # "random" and "sandbox" name the reference's built-in loaders (config.rs:39-42)
    assert host.Session(cfg, scene="random").desc.n_primitives > 400


def test_cli_flags_and_exit_codes(tmp_path):
    exe = os.path.join(ROOT, "racer-tracer_amd", "bin", "racer-tracer-amd")
    assert os.path.exists(exe)
    run = lambda *a, **kw: subprocess.run([exe] + list(a), capture_output=True, text=True, **kw)
    assert run("--help").returncode == 0
    assert run("--bogus").returncode == 10
    assert run("-c", str(tmp_path / "absent.yml")).returncode == 3
    env = dict(os.environ, CONFIG=os.path.join(SCENES, "config_c1.yml"))     # env CONFIG (config.rs:17)
    r = run("-s", "scene.txt", env=env)
    assert r.returncode == 10 and "suitable scene loader" in r.stderr
    r = run("-s", os.path.join(SCENES, "three_balls.yml"), "--image-action", "png", env=env)
    rt = importlib.import_module("racer-tracer_amd")
    if rt.device_count() == 0:
        assert r.returncode == 100 and "HIP device" in r.stderr               # loud, no CPU fallback
    else:
        assert r.returncode == 0 and "Saved image to" in r.stderr
        for f in os.listdir("."):
            if len(f) == 68 and f.endswith(".png"):
                os.remove(f)
