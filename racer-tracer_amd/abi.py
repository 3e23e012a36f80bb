"""ctypes mirror of include/rt_abi.h (struct layouts, enums, prototypes).

Plumbing only: every struct here must stay field-for-field identical to the
C header; tests/test_abi_layout.py checks the sizes against the C compiler.
"""
import ctypes as C

ABI_VERSION = 5

# RtError (reference codes: racer-tracer/src/error.rs:71-97)
RT_OK = 0
RT_ERR_CONFIGURATION = 3
RT_ERR_UNKNOWN_MATERIAL = 4
RT_ERR_CANCEL_EVENT = 7
RT_ERR_IMAGE_SAVE = 8
RT_ERR_SCENE_LOAD = 9
RT_ERR_ARGUMENT_PARSING = 10
RT_ERR_FAILED_TO_OPEN_IMAGE = 21
RT_ERR_NO_DEVICE = 100
RT_ERR_HIP = 101
RT_ERR_INVALID_ARGUMENT = 102
RT_ERR_UNSUPPORTED = 103
RT_ERR_OUT_OF_MEMORY = 104

RT_TEX_SOLID_COLOR, RT_TEX_CHECKERED, RT_TEX_IMAGE, RT_TEX_NOISE = 0, 1, 2, 3
RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE_LIGHT = 0, 1, 2, 3
RT_PRIM_SPHERE, RT_PRIM_XY_RECT, RT_PRIM_XZ_RECT, RT_PRIM_YZ_RECT, RT_PRIM_BOX = 0, 1, 2, 3, 4
RT_PRIM_MOVING_SPHERE = 5
RT_PRIM_HAS_ROTATE_Y, RT_PRIM_HAS_TRANSLATE = 1, 2
RT_BG_SKY, RT_BG_SOLID = 0, 1

D3 = C.c_double * 3


class RtTexture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("tex_even", C.c_int32), ("tex_odd", C.c_int32),
                ("image", C.c_int32), ("perlin", C.c_int32), ("depth", C.c_int32),
                ("color", D3), ("scale", C.c_double)]


class RtImage(C.Structure):
    _fields_ = [("rgba", C.POINTER(C.c_uint8)), ("width", C.c_int32), ("height", C.c_int32)]


class RtPerlin(C.Structure):
    _fields_ = [("ranvec", (C.c_double * 3) * 256), ("perm_x", C.c_int32 * 256),
                ("perm_y", C.c_int32 * 256), ("perm_z", C.c_int32 * 256)]


class RtMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("fuzz", C.c_double),
                ("refraction_index", C.c_double)]


class RtPrimitive(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("flags", C.c_int32),
                ("obj_id", C.c_int32), ("p", C.c_double * 6), ("rot_sin", C.c_double),
                ("rot_cos", C.c_double), ("translate", D3), ("center_b", D3), ("time_a", C.c_double),
                ("time_b", C.c_double)]


class RtBackground(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("top", D3), ("bottom", D3)]


class RtSceneDesc(C.Structure):
    _fields_ = [("primitives", C.POINTER(RtPrimitive)), ("n_primitives", C.c_int32),
                ("materials", C.POINTER(RtMaterial)), ("n_materials", C.c_int32),
                ("textures", C.POINTER(RtTexture)), ("n_textures", C.c_int32),
                ("images", C.POINTER(RtImage)), ("n_images", C.c_int32),
                ("perlins", C.POINTER(RtPerlin)), ("n_perlins", C.c_int32),
                ("background", RtBackground)]


class RtCamera(C.Structure):
    _fields_ = [("origin", D3), ("upper_left_corner", D3), ("forward", D3), ("right", D3),
                ("up", D3), ("horizontal", D3), ("vertical", D3), ("vfov", C.c_double),
                ("viewport_width", C.c_double), ("viewport_height", C.c_double),
                ("lens_radius", C.c_double), ("focus_distance", C.c_double),
                ("time_a", C.c_double), ("time_b", C.c_double)]


class RtRenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32),
                ("max_depth", C.c_int32), ("tiles_w", C.c_int32), ("tiles_h", C.c_int32),
                ("seed", C.c_uint64), ("strip_rows", C.c_int32), ("strip_count", C.c_int32),
                ("strip_index", C.c_int32), ("scale", C.c_int32)]


class RtRenderStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("kernel_ms", C.c_double),
                ("resolve_ms", C.c_double), ("kernel_launches", C.c_int32), ("_pad", C.c_int32)]


RT_TM_NONE, RT_TM_REINHARD, RT_TM_HABLE, RT_TM_ACES = 0, 1, 2, 3


class RtToneMap(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("max_white", C.c_double),
                ("hable", C.c_double * 6), ("exposure_bias", C.c_double), ("linear_white", C.c_double),
                ("aces_in", C.c_double * 9), ("aces_out", C.c_double * 9)]


RT_HIT_AUTO, RT_HIT_LINEAR, RT_HIT_BVH = 0, 1, 2
RT_KERNEL_POOL, RT_KERNEL_V1 = 0, 1
RT_ARITH_FAST, RT_ARITH_REFERENCE = 0, 1
RT_GATHER_AUTO, RT_GATHER_STAGED = 0, 1


class RtSceneOptions(C.Structure):
    _fields_ = [("closest_hit", C.c_int32), ("kernel", C.c_int32), ("arithmetic", C.c_int32), ("gather", C.c_int32),
                ("_reserved", C.c_int32 * 4)]


RtTileCallback = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_int32,
                             C.c_int32, C.c_int32)
# int (*RtCancelCallback)(void *cancel_user): non-zero = stop (renderer.rs:25-30 do_cancel)
RtCancelCallback = C.CFUNCTYPE(C.c_int, C.c_void_p)

# Every symbol include/rt_abi.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "rt_abi_version": (C.c_int, []),
    "rt_device_count": (C.c_int, []),
    "rt_scene_create": (C.c_int, [C.POINTER(RtSceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "rt_scene_create_ex": (C.c_int, [C.POINTER(RtSceneDesc), C.c_int, C.POINTER(RtSceneOptions), C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_release_cached_buffers": (None, []),
    "rt_render_frame": (C.c_int, [C.c_void_p, C.POINTER(RtCamera), C.POINTER(RtRenderParams),
                                  C.POINTER(C.c_double)]),
    "rt_render_frame_device": (C.c_int, [C.c_void_p, C.POINTER(RtCamera),
                                         C.POINTER(RtRenderParams), C.c_void_p, C.c_void_p]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(RtCamera), C.POINTER(RtRenderParams),
                            RtTileCallback, C.c_void_p, C.POINTER(C.c_int)]),
    "rt_render_ex": (C.c_int, [C.c_void_p, C.POINTER(RtCamera), C.POINTER(RtRenderParams),
                               RtTileCallback, C.c_void_p, RtCancelCallback, C.c_void_p]),
    "rt_render_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(RtCamera), C.POINTER(RtRenderParams), C.c_int,
                                  RtTileCallback, C.c_void_p, RtCancelCallback, C.c_void_p]),
    "rt_post_rgba8_device": (C.c_int, [C.c_void_p, C.POINTER(RtToneMap), C.c_void_p, C.c_size_t, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "rt_render_frame_rgba8": (C.c_int, [C.c_void_p, C.POINTER(RtCamera), C.POINTER(RtRenderParams),
                                        C.POINTER(RtToneMap), C.POINTER(C.c_uint8)]),
    "rt_render_frame_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(RtCamera),
                                        C.POINTER(RtRenderParams), C.c_int, C.POINTER(C.c_double)]),
    "rt_render_frame_multi_device": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(RtCamera),
                                               C.POINTER(RtRenderParams), C.c_int, C.c_void_p]),
    "rt_scene_last_stats": (C.c_int, [C.c_void_p, C.POINTER(RtRenderStats)]),
    "rt_strerror": (C.c_char_p, [C.c_int]),
    "rt_last_error_message": (C.c_char_p, []),
}


def bind(lib, prototypes):
    """Attach restype/argtypes; raises AttributeError if a symbol is missing."""
    for name, (res, args) in prototypes.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


class SceneBundle:
    """An RtSceneDesc plus the Python objects that own its memory."""

    def __init__(self, primitives, materials, textures, background, images=(), perlins=()):
        self.primitives = (RtPrimitive * max(1, len(primitives)))(*primitives)
        self.materials = (RtMaterial * max(1, len(materials)))(*materials)
        self.textures = (RtTexture * max(1, len(textures)))(*textures)
        self._image_arrays = []  # numpy uint8 arrays kept alive
        imgs = []
        for arr in images:  # arr: numpy uint8 [h, w, 4], C-contiguous
            self._image_arrays.append(arr)
            imgs.append(RtImage(arr.ctypes.data_as(C.POINTER(C.c_uint8)), arr.shape[1], arr.shape[0]))
        self.images = (RtImage * max(1, len(imgs)))(*imgs)
        self.perlins = (RtPerlin * max(1, len(perlins)))(*perlins)
        self.desc = RtSceneDesc(self.primitives, len(primitives), self.materials, len(materials),
                                self.textures, len(textures), self.images, len(imgs),
                                self.perlins, len(perlins), background)


def sphere(center, radius, material, obj_id=0):
    return RtPrimitive(RT_PRIM_SPHERE, material, 0, obj_id,
                       (C.c_double * 6)(center[0], center[1], center[2], radius, 0.0, 0.0),
                       0.0, 1.0, D3(0, 0, 0))


def moving_sphere(center_a, center_b, radius, material, obj_id=0, time_a=0.0, time_b=1.0):
    return RtPrimitive(RT_PRIM_MOVING_SPHERE, material, 0, obj_id,
                       (C.c_double * 6)(center_a[0], center_a[1], center_a[2], radius, 0.0, 0.0),
                       0.0, 1.0, D3(0, 0, 0), D3(*center_b), time_a, time_b)


def rect(kind, a0, a1, b0, b1, k, material, obj_id=0):
    return RtPrimitive(kind, material, 0, obj_id, (C.c_double * 6)(a0, a1, b0, b1, k, 0.0),
                       0.0, 1.0, D3(0, 0, 0))


def box(mn, mx, material, obj_id=0):
    return RtPrimitive(RT_PRIM_BOX, material, 0, obj_id,
                       (C.c_double * 6)(mn[0], mn[1], mn[2], mx[0], mx[1], mx[2]),
                       0.0, 1.0, D3(0, 0, 0))


def solid(color):
    return RtTexture(RT_TEX_SOLID_COLOR, -1, -1, -1, -1, 0, D3(*color), 0.0)


def material(kind, texture=-1, fuzz=0.0, ior=0.0):
    return RtMaterial(kind, texture, fuzz, ior)


def sky(top=(1.0, 1.0, 1.0), bottom=(0.5, 0.7, 1.0)):
    return RtBackground(RT_BG_SKY, 0, D3(*top), D3(*bottom))


def solid_background(color):
    return RtBackground(RT_BG_SOLID, 0, D3(*color), D3(0, 0, 0))


def render_params(width, height, samples, max_depth=20, tiles_w=10, tiles_h=10, seed=1,
                  strip_rows=0, strip_count=0, strip_index=0, scale=0):
    return RtRenderParams(width, height, samples, max_depth, tiles_w, tiles_h, seed,
                          strip_rows, strip_count, strip_index, scale)
