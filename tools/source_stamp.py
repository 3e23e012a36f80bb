#!/usr/bin/env python3
"""SHA-256 over the sources the device kernels are built from.

A PMC summary under profiles/ is only evidence for the kernel it was collected
on: tools/gpu_pmc.sh stores this stamp next to the counters, and bench.py prints
the counter-derived roofline figures only when the stamp still matches the
sources in the tree (otherwise it prints null and says why).
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


def kernel_source_sha():
    h = hashlib.sha256()
    for path in kernel_source_files():
        h.update(os.path.relpath(path, ROOT).encode())
        h.update(b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_source_sha())
