"""The device BVH builder (racer-tracer_amd/csrc/rt_bvh.cpp) is host code: tools/sim/bvh_sim.cpp walks the tree it
builds for the `random` scene on the CPU — by skip links (the device walk), by skip links two nodes at a time and in
near-to-far order with a stack — over a frame of primary rays, two bounces and a set of axis-parallel rays, and
compares every closest hit with a linear scan over the primitives.  The box test is the device's own code
(csrc/rt_bvh_slab.h, f32 fma around the root's centre), compiled for the host."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "racer-tracer_amd")


@pytest.fixture(scope="module")
def bvh_sim():
    obj = os.path.join(PKG, "build", "product", "rt_bvh.o")
    lib = os.path.join(PKG, "lib", "libracer_tracer_amd.so")
    if not (os.path.exists(obj) and os.path.exists(lib)):
        pytest.skip("the library has not been built (python -c 'import __graft_entry__ as g; g.build()')")
    exe = os.path.join(PKG, "build", "bvh_sim")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + PKG, "-o", exe,
                    os.path.join(ROOT, "tools", "sim", "bvh_sim.cpp"), obj, "-L" + os.path.join(PKG, "lib"),
                    "-lracer_tracer_amd", "-Wl,-rpath," + os.path.join(PKG, "lib")], check=True, cwd=ROOT)
    return exe


@pytest.mark.parametrize("max_leaf", [1, 3, 4, 7])
def test_every_walk_finds_the_linear_scan_hit(bvh_sim, max_leaf):
    out = subprocess.run([bvh_sim, str(max_leaf)], check=True, cwd=ROOT, capture_output=True, text=True).stdout
    head = re.search(r"(\d+) primitives, (\d+) nodes \((\d+) leaves, (\d+) inner\), depth (\d+)", out)
    n_prims, n_nodes, leaves, inner, depth = map(int, head.groups())
    assert n_prims == 485 and leaves == inner + 1 and n_nodes == leaves + inner      # a full binary tree
    assert leaves >= (n_prims + max_leaf - 1) // max_leaf and depth <= 24
    # axis-parallel rays (direction components of exactly 0, 1/d beyond the f32 range) through the DEVICE's f32 slab test
    axis = re.search(r"axis-parallel: (\d+) rays, (\d+) hit something \| closest hits differ: (\d+)", out)
    assert int(axis.group(1)) >= 10000 and int(axis.group(2)) > 1000 and int(axis.group(3)) == 0
    bounces = re.findall(r"bounce (\d): (\d+) rays .* closest hits differ: (\d+)", out)
    assert [b[0] for b in bounces] == ["0", "1", "2"]
    for _, rays, differ in bounces:
        assert int(rays) > 10000 and int(differ) == 0


def test_camera_ordered_node_array_walks_to_the_same_hits_with_fewer_boxes():
    """rt_bvh.h: order_bvh_for_origin — the array enqueue_render uploads for trees that live in LDS, every node's children
    nearest-to-the-camera first.  The skip-link walk over it must find the linear scan's closest hit for every primary and
    bounce ray of the `random` scene, and primary rays must touch clearly fewer boxes than over the builder's own order."""
    obj = os.path.join(PKG, "build", "product", "rt_bvh.o")
    lib = os.path.join(PKG, "lib", "libracer_tracer_amd.so")
    if not (os.path.exists(obj) and os.path.exists(lib)):
        pytest.skip("the library has not been built (python -c 'import __graft_entry__ as g; g.build()')")
    exe = os.path.join(PKG, "build", "bvh_rotate")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + PKG, "-o", exe,
                    os.path.join(ROOT, "tools", "sim", "bvh_rotate.cpp"), obj, "-L" + os.path.join(PKG, "lib"),
                    "-lracer_tracer_amd", "-Wl,-rpath," + os.path.join(PKG, "lib")], check=True, cwd=ROOT)
    out = subprocess.run([exe, "3"], check=True, cwd=ROOT, capture_output=True, text=True, timeout=600).stdout
    m = re.search(r"order_bvh_for_origin\(camera\): (\d+) rays whose closest hit differs.*boxes per primary ray ([\d.]+) -> ([\d.]+), per bounce ray ([\d.]+) -> ([\d.]+)", out)
    assert m, out[-400:]
    mismatches, p0, p1, b0, b1 = int(m.group(1)), *map(float, m.groups()[1:])
    assert mismatches == 0
    assert p1 < 0.9 * p0 and abs(b1 - b0) < 0.05 * b0
