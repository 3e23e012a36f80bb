"""racer-tracer_amd — ctypes binding of libracer_tracer_amd.so (MI355X / gfx950).

The package name contains a hyphen like the reference crate's, so import it
with ``importlib.import_module("racer-tracer_amd")``.

This is plumbing over the C ABI of include/rt_abi.h (device path) and
include/rt_host.h (scene/config loading, tone map, PNG).  There is NO CPU
fallback: if the shared library is missing or no GPU is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# RACER_TRACER_AMD_LIB lets a developer point at another build of the same ABI
LIB_PATH = os.environ.get("RACER_TRACER_AMD_LIB", os.path.join(_HERE, "lib", "libracer_tracer_amd.so"))


class RtError(RuntimeError):
    def __init__(self, code, what, detail):
        super().__init__("%s failed: [%d] %s%s" % (what, code, _strerror(code), (": " + detail) if detail else ""))
        self.code = code


_lib = None


def lib():
    """The loaded C-ABI library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `make -C %s` (or __graft_entry__.build()); "
                "there is no CPU fallback for the render path" % (LIB_PATH, _HERE))
        # torch's ROCm wheels carry their own libamdhip64; whichever HIP runtime a process loads first
        # serves everything after it, and torch cannot see the GPU through /opt/rocm's copy if this
        # library pulled that in first (the other order is fine).  So a process that has torch loads
        # torch's runtime first; one without torch (the C++/Rust hosts, the CLI) never notices.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = abi.bind(C.CDLL(LIB_PATH), abi.PROTOTYPES)
        if _lib.rt_abi_version() != abi.ABI_VERSION:
            raise ImportError("ABI version mismatch: library %d, binding %d"
                              % (_lib.rt_abi_version(), abi.ABI_VERSION))
    return _lib


def load_library(path):
    """Another build of the same ABI (e.g. build/libracer_tracer_amd_exact.so, the tests' exact-arithmetic
    build) next to the default one: pass the result as Scene(..., library=...)."""
    lib()  # the default library (and torch's HIP runtime) first
    other = abi.bind(C.CDLL(path), abi.PROTOTYPES)
    if other.rt_abi_version() != abi.ABI_VERSION:
        raise ImportError("ABI version mismatch in %s" % path)
    return other


def _strerror(code):
    try:
        return lib().rt_strerror(code).decode()
    except Exception:  # pragma: no cover - only when the library itself is missing
        return "?"


def check(code, what, library=None):
    if code != abi.RT_OK:
        raise RtError(code, what, (library or lib()).rt_last_error_message().decode())


def device_count():
    return lib().rt_device_count()


def render_frame_multi(scenes, camera, params, strip_rows=0):
    """rt_render_frame_multi: one frame over several Scene objects (one per device share)
    -> float64 [H, W, 3] on the host."""
    out = np.zeros((params.height, params.width, 3), dtype=np.float64)
    handles = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    check(lib().rt_render_frame_multi(handles, len(scenes), C.byref(camera), C.byref(params), strip_rows,
                                      out.ctypes.data_as(C.POINTER(C.c_double))), "rt_render_frame_multi")
    return out


def render_frame_multi_device(scenes, camera, params, out_ptr, strip_rows=0):
    """rt_render_frame_multi_device: out_ptr = device address (int) on scenes[0]'s device."""
    handles = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    check(lib().rt_render_frame_multi_device(handles, len(scenes), C.byref(camera), C.byref(params), strip_rows,
                                             C.c_void_p(out_ptr)), "rt_render_frame_multi_device")


def _tile_collector(tiles):
    def on_tile(_user, rgb, r, c, w, h):
        # `rgb` is only valid during the callback; an empty tile (more tile rows / columns than pixels) carries no pixels
        arr = np.ctypeslib.as_array(rgb, shape=(h, w, 3)).copy() if w > 0 and h > 0 else np.zeros((h, w, 3))
        tiles.append((r, c, w, h, arr))
    return on_tile


def render_tiles_multi(scenes, camera, params, strip_rows=0, cancel=None):
    """rt_render_multi: the tile stream of one frame rendered by several Scene objects (one per device share)
    -> list of (r, c, width, height, float64 [height, width, 3]); cancel: None or a callable."""
    tiles = []
    cb = abi.RtTileCallback(_tile_collector(tiles))
    hook = abi.RtCancelCallback(lambda _user: 1 if cancel() else 0) if cancel is not None else C.cast(None, abi.RtCancelCallback)
    handles = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    check(lib().rt_render_multi(handles, len(scenes), C.byref(camera), C.byref(params), strip_rows, cb, None, hook, None),
          "rt_render_multi")
    return tiles


class Scene:
    """RtScene handle: a scene uploaded to one GPU (rt_scene_create)."""

    def __init__(self, desc, device=0, closest_hit=abi.RT_HIT_AUTO, kernel=abi.RT_KERNEL_POOL, library=None,
                 arithmetic=abi.RT_ARITH_FAST, gather=abi.RT_GATHER_AUTO):
        """closest_hit / kernel / arithmetic / gather: RtSceneOptions (rt_scene_create_ex) — the defaults are
        rt_scene_create's own; arithmetic=RT_ARITH_REFERENCE selects the reference's IEEE divisions without FMA
        contraction; gather=RT_GATHER_STAGED makes rt_render_frame_multi_device stage and copy this scene's strips
        even on the output's device.  library: load_library()."""
        self._lib = library or lib()
        self._h = C.c_void_p()
        self._desc_owner = desc  # keep SceneBundle / host session alive
        d = desc.desc if hasattr(desc, "desc") else desc
        opt = abi.RtSceneOptions(closest_hit, kernel, arithmetic, gather)
        check(self._lib.rt_scene_create_ex(C.byref(d), device, C.byref(opt), C.byref(self._h)), "rt_scene_create_ex", self._lib)
        self.device = device

    def close(self):
        if self._h:
            self._lib.rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_frame(self, camera, params):
        """rt_render_frame -> float64 [H, W, 3], gamma-encoded, not tone-mapped."""
        out = np.zeros((params.height, params.width, 3), dtype=np.float64)
        check(self._lib.rt_render_frame(self._h, C.byref(camera), C.byref(params),
                                    out.ctypes.data_as(C.POINTER(C.c_double))), "rt_render_frame", self._lib)
        return out

    def render_frame_device(self, camera, params, out_ptr, stream=None):
        """rt_render_frame_device: out_ptr = device address (int), stream = hipStream_t (int)."""
        check(self._lib.rt_render_frame_device(self._h, C.byref(camera), C.byref(params),
                                           C.c_void_p(out_ptr), C.c_void_p(stream or 0)),
              "rt_render_frame_device", self._lib)

    def render_frame_rgba8(self, camera, params, tone_map):
        """rt_render_frame_rgba8 -> uint8 [H, W, 4]: render, tone-map and pack on the device."""
        out = np.zeros((params.height, params.width, 4), dtype=np.uint8)
        check(self._lib.rt_render_frame_rgba8(self._h, C.byref(camera), C.byref(params), C.byref(tone_map),
                                          out.ctypes.data_as(C.POINTER(C.c_uint8))), "rt_render_frame_rgba8", self._lib)
        return out

    def post_rgba8_device(self, tone_map, rgb_ptr, n_pixels, rgba_ptr, mapped_ptr=None, stream=None):
        """rt_post_rgba8_device on device addresses (ints)."""
        check(self._lib.rt_post_rgba8_device(self._h, C.byref(tone_map), C.c_void_p(rgb_ptr), n_pixels,
                                         C.c_void_p(rgba_ptr), C.c_void_p(mapped_ptr or 0), C.c_void_p(stream or 0)),
              "rt_post_rgba8_device", self._lib)

    def render_tiles(self, camera, params, cancel=None):
        """rt_render / rt_render_ex -> list of (r, c, width, height, float64 [height, width, 3]).
        cancel: None, a ctypes pointer to an int flag (rt_render), or a callable returning True once the
        render should stop (rt_render_ex's RtCancelCallback — how the reference's SignalEvent binds)."""
        tiles = []
        cb = abi.RtTileCallback(_tile_collector(tiles))
        if callable(cancel):
            hook = abi.RtCancelCallback(lambda _user: 1 if cancel() else 0)
            check(self._lib.rt_render_ex(self._h, C.byref(camera), C.byref(params), cb, None, hook, None), "rt_render_ex", self._lib)
        else:
            cancel_ptr = C.cast(cancel, C.POINTER(C.c_int)) if cancel is not None else None
            check(self._lib.rt_render(self._h, C.byref(camera), C.byref(params), cb, None, cancel_ptr), "rt_render", self._lib)
        return tiles

    def last_stats(self):
        st = abi.RtRenderStats()
        check(self._lib.rt_scene_last_stats(self._h, C.byref(st)), "rt_scene_last_stats", self._lib)
        return st
