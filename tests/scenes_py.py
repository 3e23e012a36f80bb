"""Hand-transcribed POD forms of the reference's shipped scenes.

Independent of the C++ YAML loader on purpose: tests compare the loader's
output against these, and the oracle pinning tests use these directly.
Values are the DATA of /root/reference/resources/scenes/*.yml (object order =
file order; the reference's HashMap order is random, scene/yml.rs:442).
"""
import importlib
import math
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
abi = importlib.import_module("racer-tracer_amd.abi")

L, M, D, E = abi.RT_MAT_LAMBERTIAN, abi.RT_MAT_METAL, abi.RT_MAT_DIELECTRIC, abi.RT_MAT_DIFFUSE_LIGHT


def three_balls():
    """three_balls.yml: 5 spheres, default Sky, tone map None."""
    textures = [abi.solid((0.8, 0.8, 0.0)), abi.solid((0.1, 0.2, 0.5)), abi.solid((0.8, 0.6, 0.2))]
    materials = [abi.material(L, 0), abi.material(L, 1), abi.material(D, -1, ior=1.5),
                 abi.material(M, 2, fuzz=0.0)]
    prims = [abi.sphere((0.0, -100.5, -1.0), 100.0, 0, 1),
             abi.sphere((0.0, 0.0, -1.0), 0.5, 1, 2),
             abi.sphere((-1.0, 0.0, -1.0), 0.5, 2, 3),
             abi.sphere((-1.0, 0.0, -1.0), -0.4, 2, 4),
             abi.sphere((1.0, 0.0, -1.0), 0.5, 3, 5)]
    cam = dict(look_from=(0.0, 2.0, 10.0), look_at=(0.0, 0.0, 0.0), vfov=20.0, aperture=0.1,
               focus_distance=10.0)
    return abi.SceneBundle(prims, materials, textures, abi.sky()), cam, "None"


def two_balls():
    textures = [abi.solid((0.0, 0.0, 1.0)), abi.solid((1.0, 0.0, 0.0))]
    materials = [abi.material(L, 0), abi.material(L, 1)]
    prims = [abi.sphere((-0.707, 0.0, -1.0), 0.707, 0, 1), abi.sphere((0.707, 0.0, -1.0), 0.707, 1, 2)]
    cam = dict(look_from=(0.0, 2.0, 10.0), look_at=(0.0, 0.0, 0.0), vfov=20.0, aperture=0.1,
               focus_distance=10.0)
    return abi.SceneBundle(prims, materials, textures, abi.sky()), cam, "None"


def cornell_box():
    """cornell_box.yml: 5 Lambertian rects + 1 light rect, black background,
    no tone_map key (falls back to the config's Aces, main.rs:84-86)."""
    textures = [abi.solid((0.12, 0.45, 0.15)), abi.solid((0.65, 0.05, 0.05)),
                abi.solid((0.63, 0.63, 0.63)), abi.solid((15.0, 15.0, 15.0))]
    materials = [abi.material(L, 0), abi.material(L, 1), abi.material(L, 2), abi.material(E, 3)]
    prims = [abi.rect(abi.RT_PRIM_YZ_RECT, 0, 555, 0, 555, 555, 0, 1),
             abi.rect(abi.RT_PRIM_YZ_RECT, 0, 555, 0, 555, 0, 1, 2),
             abi.rect(abi.RT_PRIM_XZ_RECT, 0, 555, 0, 555, 0, 2, 3),
             abi.rect(abi.RT_PRIM_XZ_RECT, 0, 555, 0, 555, 555, 2, 4),
             abi.rect(abi.RT_PRIM_XY_RECT, 0, 555, 0, 555, 555, 2, 5),
             abi.rect(abi.RT_PRIM_XZ_RECT, 213, 343, 227, 332, 554, 3, 6)]
    cam = dict(look_from=(278.0, 278.0, -800.0), look_at=(278.0, 278.0, 0.0), vfov=40.0,
               aperture=0.0, focus_distance=10000.0)
    return abi.SceneBundle(prims, materials, textures, abi.solid_background((0.0, 0.0, 0.0))), cam, "Aces"


def cornell_box_boxes():
    """Content of the Sandbox loader (scene/sandbox.rs:39-80): cornell_box.yml
    plus two white boxes, each RotateY then Translate."""
    bundle, cam, tm = cornell_box()
    prims = list(bundle.primitives)[:6]
    for mx, deg, off, oid in (((165.0, 330.0, 165.0), 15.0, (265.0, 0.0, 295.0), 7),
                              ((165.0, 165.0, 165.0), -18.0, (130.0, 0.0, 65.0), 8)):
        b = abi.box((0.0, 0.0, 0.0), mx, 2, oid)
        rad = deg * math.pi / 180.0
        b.flags = abi.RT_PRIM_HAS_ROTATE_Y | abi.RT_PRIM_HAS_TRANSLATE
        b.rot_sin, b.rot_cos = math.sin(rad), math.cos(rad)
        b.translate = abi.D3(*off)
        prims.append(b)
    out = abi.SceneBundle(prims, list(bundle.materials)[:4], list(bundle.textures)[:4],
                          abi.solid_background((0.0, 0.0, 0.0)))
    return out, cam, "Aces"


def camera_for(cam, width, height):
    from oracle import oracle_ctypes as orc
    return orc.camera(cam["look_from"], cam["look_at"], cam["vfov"], cam["aperture"],
                      cam["focus_distance"], width, height)
