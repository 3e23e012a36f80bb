"""ctypes binding of include/rt_host.h: config + scene loading, tone map, PNG.

Plumbing over the C++ host layer inside libracer_tracer_amd.so.
"""
import ctypes as C

import numpy as np

from . import abi, check as _check_rt, lib as _lib

_P = C.POINTER
_PROTOS = {
    "rth_session_open": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64, _P(C.c_void_p)]),
    "rth_session_close": (None, [C.c_void_p]),
    "rth_session_scene": (_P(abi.RtSceneDesc), [C.c_void_p]),
    "rth_session_camera": (_P(abi.RtCamera), [C.c_void_p]),
    "rth_session_params": (C.c_int, [C.c_void_p, C.c_int, _P(abi.RtRenderParams)]),
    "rth_session_image_action": (C.c_int, [C.c_void_p]),
    "rth_session_tone_map_kind": (C.c_int, [C.c_void_p]),
    "rth_session_image_output_dir": (C.c_char_p, [C.c_void_p]),
    "rth_session_tone_map": (C.c_int, [C.c_void_p, _P(abi.RtToneMap)]),
    "rth_tone_map": (C.c_int, [C.c_void_p, _P(C.c_double), _P(C.c_double), C.c_size_t]),
    "rth_pack_rgba8": (C.c_int, [_P(C.c_double), C.c_size_t, _P(C.c_uint8)]),
    "rth_save_png": (C.c_int, [C.c_void_p, _P(C.c_double), C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]),
    "rth_camera_new": (C.c_int, [_P(C.c_double), _P(C.c_double), C.c_double, C.c_double, C.c_double,
                                 C.c_int, C.c_int, _P(abi.RtCamera)]),
    "rth_decode_image": (C.c_int, [C.c_char_p, _P(_P(C.c_uint8)), _P(C.c_int), _P(C.c_int)]),
    "rth_free": (None, [C.c_void_p]),
    "rth_sha256_hex": (C.c_int, [_P(C.c_uint8), C.c_size_t, C.c_char_p]),
    "rth_last_error_message": (C.c_char_p, []),
}

TONE_MAP_NAMES = {0: "None", 1: "Reinhard", 2: "Hable", 3: "Aces"}


class HostError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed: [%d] %s" % (what, code, hlib().rth_last_error_message().decode()))
        self.code = code


_bound = None


def hlib():
    global _bound
    if _bound is None:
        _bound = abi.bind(_lib(), _PROTOS)
    return _bound


def _check(code, what):
    if code != abi.RT_OK:
        raise HostError(code, what)


def _b(s):
    return None if s is None else str(s).encode()


class Session:
    """rth_session_open: config.yml (+ overrides) -> scene PODs, camera, params."""

    def __init__(self, config_path, scene=None, image_action=None, seed=1):
        self._h = C.c_void_p()
        _check(hlib().rth_session_open(_b(config_path), _b(scene), _b(image_action), seed, C.byref(self._h)),
               "rth_session_open")
        self.desc = hlib().rth_session_scene(self._h).contents
        self.camera = hlib().rth_session_camera(self._h).contents
        self.params = abi.RtRenderParams()
        _check(hlib().rth_session_params(self._h, 0, C.byref(self.params)), "rth_session_params")
        self.preview_params = abi.RtRenderParams()
        _check(hlib().rth_session_params(self._h, 1, C.byref(self.preview_params)), "rth_session_params")
        self.image_action = hlib().rth_session_image_action(self._h)
        self.tone_map_name = TONE_MAP_NAMES[hlib().rth_session_tone_map_kind(self._h)]
        self.tone_map_desc = abi.RtToneMap()
        _check(hlib().rth_session_tone_map(self._h, C.byref(self.tone_map_desc)), "rth_session_tone_map")
        d = hlib().rth_session_image_output_dir(self._h)
        self.image_output_dir = d.decode() if d is not None else None

    def close(self):
        if self._h:
            hlib().rth_session_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tone_map(self, rgb):
        src = np.ascontiguousarray(rgb, dtype=np.float64)
        dst = np.empty_like(src)
        _check(hlib().rth_tone_map(self._h, src.ctypes.data_as(_P(C.c_double)),
                                   dst.ctypes.data_as(_P(C.c_double)), src.size // 3), "rth_tone_map")
        return dst

    def save_png(self, tone_mapped_rgb, directory=None):
        src = np.ascontiguousarray(tone_mapped_rgb, dtype=np.float64)
        h, w = src.shape[0], src.shape[1]
        buf = C.create_string_buffer(4096)
        _check(hlib().rth_save_png(self._h, src.ctypes.data_as(_P(C.c_double)), w, h, _b(directory), buf, 4096),
               "rth_save_png")
        return buf.value.decode()


def pack_rgba8(rgb):
    src = np.ascontiguousarray(rgb, dtype=np.float64)
    out = np.empty(src.shape[:-1] + (4,), dtype=np.uint8)
    _check(hlib().rth_pack_rgba8(src.ctypes.data_as(_P(C.c_double)), src.size // 3,
                                 out.ctypes.data_as(_P(C.c_uint8))), "rth_pack_rgba8")
    return out


def camera_new(look_from, look_at, vfov, aperture, focus_distance, width, height):
    cam = abi.RtCamera()
    _check(hlib().rth_camera_new(abi.D3(*look_from), abi.D3(*look_at), vfov, aperture, focus_distance,
                                 width, height, C.byref(cam)), "rth_camera_new")
    return cam


def decode_image(path):
    p = _P(C.c_uint8)()
    w, h = C.c_int(), C.c_int()
    _check(hlib().rth_decode_image(_b(path), C.byref(p), C.byref(w), C.byref(h)), "rth_decode_image")
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    finally:
        hlib().rth_free(p)


def sha256_hex(data):
    buf = C.create_string_buffer(65)
    arr = (C.c_uint8 * len(data)).from_buffer_copy(data) if data else None
    _check(hlib().rth_sha256_hex(arr, len(data), buf), "rth_sha256_hex")
    return buf.value.decode()
