"""The C-ABI library loads on a CPU-only machine, exports every symbol the
headers declare, and the ctypes mirrors match the C layouts (checked against
gcc).  No compute call is made here."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def declared_functions(header):
    text = open(os.path.join(INC, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if not n.endswith("_t")))


def test_library_exports_every_declared_symbol(rt):
    lib = C.CDLL(rt.LIB_PATH)
    missing = [n for h in ("rt_abi.h", "rt_host.h") for n in declared_functions(h) if not hasattr(lib, n)]
    assert not missing, missing
    assert len(declared_functions("rt_abi.h")) >= 10 and len(declared_functions("rt_host.h")) >= 15


def test_python_prototypes_cover_the_headers(rt):
    host = __import__("importlib").import_module("racer-tracer_amd.host")
    assert sorted(rt.abi.PROTOTYPES) == declared_functions("rt_abi.h")
    assert sorted(host._PROTOS) == declared_functions("rt_host.h")


def test_abi_version_and_strerror(rt):
    assert rt.lib().rt_abi_version() == rt.abi.ABI_VERSION == 5
    # error.rs:71-97 numbering
    for code, text in ((0, "Ok"), (4, "Unknown Material"), (7, "Cancel event"), (9, "Scene failed to load"),
                       (21, "Failed to open image"), (100, "No usable HIP device")):
        assert rt.lib().rt_strerror(code).decode() == text
    assert rt.lib().rt_strerror(12345).decode() == "Unknown error"
    assert rt.device_count() >= 0


def test_struct_layouts_match_the_c_compiler(abi):
    structs = ["RtTexture", "RtImage", "RtPerlin", "RtMaterial", "RtPrimitive", "RtBackground", "RtSceneDesc",
               "RtCamera", "RtRenderParams", "RtRenderStats", "RtToneMap", "RtSceneOptions"]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rt_abi.h"', 'int main(void){']
    for s in structs:
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (s, s))
        for name, _ in getattr(abi, s)._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (s, name, s, name))
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "l.c"), os.path.join(d, "l")
        open(src, "w").write("\n".join(lines))
        subprocess.check_call(["gcc", "-I", INC, "-o", exe, src])
        out = subprocess.check_output([exe]).decode().split("\n")
    want = dict(l.split() for l in out if l)
    for s in structs:
        cls = getattr(abi, s)
        assert C.sizeof(cls) == int(want[s]), s
        for name, _ in cls._fields_:
            assert getattr(cls, name).offset == int(want["%s.%s" % (s, name)]), (s, name)


def test_headers_are_plain_c(abi):
    """The boundary must be bindable from any language: compile the headers as
    C89-ish C with no C++ and no HIP/torch includes."""
    for h in ("rt_abi.h", "rt_host.h", "rt_rng.h"):
        text = open(os.path.join(INC, h)).read()
        assert "torch" not in text and "hip/" not in text
        flags = ["-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c"]
        if h != "rt_rng.h":  # macros only: an empty translation unit is not pedantic C
            flags.insert(1, "-pedantic")
        subprocess.check_call(["gcc"] + flags + [os.path.join(INC, h)])


def test_product_library_reads_no_environment_knob():
    """Developer knobs (RT_DBG*, RT_POOL_CHUNK, ...) exist only under -DRT_DEVELOPER_KNOBS; the shipped
    library must behave the same in every environment (chunk boundaries are part of the bit-exact contract).
    The one getenv left is the reference CLI's own `CONFIG` variable (config.rs:17)."""
    so = os.path.join(ROOT, "racer-tracer_amd", "lib", "libracer_tracer_amd.so")
    strings = subprocess.check_output(["strings", so]).decode().split("\n")
    knobs = [s for s in strings if re.fullmatch(r"RT_[A-Z0-9_%d]+", s)]
    assert knobs == [], knobs
    csrc = os.path.join(ROOT, "racer-tracer_amd", "csrc")
    for f in os.listdir(csrc):
        text = open(os.path.join(csrc, f)).read()
        outside = re.sub(r"#ifdef RT_DEVELOPER_KNOBS.*?#endif", "", text, flags=re.S)
        assert "getenv" not in outside, f


def test_both_arithmetics_are_in_the_one_library(rt):
    """RT_ARITH_REFERENCE is an option of the shipped library (RtSceneOptions.arithmetic), not a second build: the
    trace kernels are in it twice, and their launchers under both names."""
    so = os.path.join(ROOT, "racer-tracer_amd", "lib", "libracer_tracer_amd.so")
    symbols = subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()
    for name in ("rtdev_launch_trace_pool", "rtdev_launch_trace_pool_exact", "rtdev_pool_blocks_per_cu_exact"):
        assert re.search(r"\b%s\b" % name, symbols), name


def test_multi_device_entry_points_validate_before_touching_a_device(rt, abi):
    """rt_render_frame_multi* (cpu.rs:118-131 over GPUs): argument errors are reported without a device."""
    cam, p = abi.RtCamera(), abi.render_params(16, 16, 1)
    out = (C.c_double * (16 * 16 * 3))()
    assert rt.lib().rt_render_frame_multi(None, 0, C.byref(cam), C.byref(p), 0, out) == abi.RT_ERR_INVALID_ARGUMENT
    handles = (C.c_void_p * 2)(None, None)
    assert rt.lib().rt_render_frame_multi(handles, 2, C.byref(cam), C.byref(p), 0, out) == abi.RT_ERR_INVALID_ARGUMENT
    assert b"NULL" in rt.lib().rt_last_error_message()
    noop = abi.RtTileCallback(lambda *a: None)
    never = C.cast(None, abi.RtCancelCallback)
    assert rt.lib().rt_render_multi(handles, 2, C.byref(cam), C.byref(p), 0, noop, None, never, None) == abi.RT_ERR_INVALID_ARGUMENT
    assert rt.lib().rt_render_ex(None, C.byref(cam), C.byref(p), noop, None, never, None) == abi.RT_ERR_INVALID_ARGUMENT


def test_scene_options_are_validated(rt, abi):
    import scenes_py as S
    bundle, _, _ = S.cornell_box()
    h = C.c_void_p()
    bad = abi.RtSceneOptions(7, 0)
    assert rt.lib().rt_scene_create_ex(C.byref(bundle.desc), 0, C.byref(bad), C.byref(h)) == abi.RT_ERR_INVALID_ARGUMENT
    bad = abi.RtSceneOptions(0, 0)
    bad._reserved[2] = 1
    assert rt.lib().rt_scene_create_ex(C.byref(bundle.desc), 0, C.byref(bad), C.byref(h)) == abi.RT_ERR_INVALID_ARGUMENT
    bad = abi.RtSceneOptions(0, 0, 2)   # RtArithmetic: FAST = 0, REFERENCE = 1
    assert rt.lib().rt_scene_create_ex(C.byref(bundle.desc), 0, C.byref(bad), C.byref(h)) == abi.RT_ERR_INVALID_ARGUMENT


def test_no_device_is_an_error_not_a_fallback(rt):
    if rt.device_count() > 0:
        pytest.skip("a GPU is visible")
    import scenes_py as S
    bundle, _, _ = S.cornell_box()
    with pytest.raises(rt.RtError) as e:
        rt.Scene(bundle)
    assert e.value.code == rt.abi.RT_ERR_NO_DEVICE


def test_product_never_touches_the_oracle():
    """The shipped path must not include, link or import anything under oracle/."""
    pkg = os.path.join(ROOT, "racer-tracer_amd")
    bad = []
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                # includes, imports, link flags or calls — mentions in comments are fine
                if re.search(r'#\s*include\s*[<"][^>"]*oracle|^\s*(from|import)\s+[\w.]*oracle|liboracle|\borc_\w+\s*\(',
                             text, flags=re.M):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
    so = os.path.join(pkg, "lib", "libracer_tracer_amd.so")
    needed = subprocess.check_output(["readelf", "-d", so]).decode()
    assert "oracle" not in needed
