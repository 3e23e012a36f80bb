#!/usr/bin/env python3
"""SHA-256 over the CODE of the sources the device kernels are built from — comments and white space excluded.

A PMC summary under profiles/ is only evidence for the kernel it was collected on: tools/pmc.py stores this stamp next
to the counters, and bench.py uses a committed summary only while the stamp still matches the sources in the tree.
Comments and layout do not reach the code object, so they do not reach the stamp either (round 2 re-collected counters
forty times for comment edits).
"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_files():
    csrc = os.path.join(ROOT, "racer-tracer_amd", "csrc")
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h", ".cpp"))]
    files.append(os.path.join(ROOT, "include", "rt_rng.h"))
    files.append(os.path.join(ROOT, "include", "rt_abi.h"))
    return files


def strip_comments(text):
    """C/C++ source without // and /* */ comments (string and character literals are respected), every run of white
    space collapsed to one blank."""
    out = []
    i, n = 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def kernel_source_sha():
    h = hashlib.sha256()
    for path in kernel_source_files():
        h.update(os.path.relpath(path, ROOT).encode())
        h.update(b"\0")
        with open(path, "r", encoding="utf-8") as f:
            h.update(strip_comments(f.read()).encode())
        h.update(b"\0")
    return h.hexdigest()


def library_path():
    """The library bench.py will load: RACER_TRACER_AMD_LIB (a developer / A-B build) or the shipped one."""
    return os.environ.get("RACER_TRACER_AMD_LIB") or os.path.join(ROOT, "racer-tracer_amd", "lib", "libracer_tracer_amd.so")


def library_sha():
    """SHA-256 of the library FILE the counters are collected on (None when it is not built): the source stamp does not see
    build flags (RT_OCC_*, AB_FLAGS) or a RACER_TRACER_AMD_LIB override, this does."""
    try:
        with open(library_path(), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None


if __name__ == "__main__":
    print(kernel_source_sha())
    print(library_sha(), library_path())
