"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers are part of the
reference's CI hygiene; here they are CPU-only, the GPU pool has no device sanitizers).  oracle/*.c are compiled by gcc
with -fsanitize=address,undefined, linked with tests/oracle_sanitize_driver.cpp and the C++ host layer (the scene
loader), and the driver renders every shipped scene and `random` at 32x18x2 through the linear scan and the
reference-shaped BVH, with strips and several thread counts, and calls every known-answer entry point of oracle.h."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"]


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    if shutil.which("gcc") is None or shutil.which("g++") is None:
        pytest.skip("no gcc / g++")
    objs = []
    for src in sorted(glob.glob(os.path.join(ROOT, "oracle", "*.c"))):   # the oracle's own flags (oracle/Makefile) + the sanitizers
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        build = subprocess.run(["gcc", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wextra", "-pthread"] + SAN + ["-c", src, "-o", obj],
                               capture_output=True, text=True)
        if build.returncode != 0 and ("asan" in build.stderr or "ubsan" in build.stderr):
            pytest.skip("sanitizer runtimes not installed: " + build.stderr[-200:])
        assert build.returncode == 0, build.stderr[-2000:]
        objs.append(obj)
    host = [f for f in glob.glob(os.path.join(ROOT, "racer-tracer_amd", "host", "*.cpp")) if not f.endswith("main.cpp")]
    exe = str(tmp_path / "oracle_sanitize")
    link = subprocess.run(["g++", "-std=c++17"] + SAN + ["-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "oracle_sanitize_driver.cpp")] + host + objs + ["-lz", "-lm", "-lpthread"],
                          capture_output=True, text=True)
    if link.returncode != 0 and ("asan" in link.stderr or "ubsan" in link.stderr):
        pytest.skip("sanitizer runtimes not installed: " + link.stderr[-200:])
    assert link.returncode == 0, link.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe, ROOT], capture_output=True, text=True, env=env, timeout=600)
    out = run.stdout + run.stderr
    assert run.returncode == 0, out[-4000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out and "LeakSanitizer" not in out, out[-4000:]
    for scene in ("three_balls", "cornell_box", "noise_and_textures", "emissive", "clown", "two_balls", "cornell_box_boxes", "random"):
        assert scene + " ok:" in out, out[-2000:]
    assert "all checks passed" in out
