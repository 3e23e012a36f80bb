"""What the hand-off inside the delivering launch rests on is in the GENERATED CODE, so it is pinned there (CPU test: hipcc
cross-compiles gfx950 without a GPU).  deliver_item (csrc/rt_trace_pool_kernel.hip) passes a tile's per-chunk slices from
the waves that wrote them to the wave that lands the tile's last chunk — waves of other XCDs, whose L2s are not coherent
with each other — with device-scope RELAXED accesses and one s_waitcnt instead of a release / acquire fence per item
(12 % of C2, profiles/r03_fence_probe.txt).  That is only sound if
  * the slice stores carry sc1 (written through to the level all XCDs share) and an `s_waitcnt vmcnt(0)` stands between
    them and the wave's bump of the tile counter,
  * the slice loads of the finishing wave carry sc1 (they bypass a stale L2 line) and come after the counter's value has
    returned (the atomic's own s_waitcnt vmcnt(0)),
  * the finished pixels are released at system scope (buffer_wbl2 sc0 sc1) before the tile is counted for its region.
The reference hands tiles over through a channel (cpu.rs:64-70); this is that channel's memory order."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def pool_kernel_asm(tmp_path_factory):
    if not os.path.exists(HIPCC) and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("isa") / "pool.s")
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S",
           os.path.join(ROOT, "racer-tracer_amd", "csrc", "rt_trace_pool_kernel.hip"), "-o", out]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    return open(out).read().split("\n")


def function_body(lines, needle):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN10rtdev_fast\d+" + needle + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return [l.strip() for l in lines[start:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]


def test_deliver_item_hands_slices_over_with_sc1_and_waitcnt(pool_kernel_asm):
    body = function_body(pool_kernel_asm, "deliver_item")
    idx = lambda pred, after=0: next(i for i in range(after, len(body)) if pred(body[i]))
    # 1. the three slice stores, device scope
    stores = [i for i, l in enumerate(body) if l.startswith("global_store_dwordx2")][:3]
    assert len(stores) == 3 and all("sc1" in body[i] for i in stores), [body[i] for i in stores]
    # 2. ... acknowledged before the tile counter is bumped
    wait = idx(lambda l: l.startswith("s_waitcnt") and "vmcnt(0)" in l, stores[-1])
    bump = idx(lambda l: l.startswith("global_atomic_add"), stores[-1])
    assert stores[-1] < wait < bump, body[stores[0]:bump + 1]
    assert not any(l.startswith(("global_store", "global_load")) for l in body[wait:bump])
    # 3. the finishing wave's slice loads: device scope, behind the counter's returned value
    returned = idx(lambda l: l.startswith("s_waitcnt") and "vmcnt(0)" in l, bump)
    loads = [i for i, l in enumerate(body) if l.startswith("global_load_dwordx2")]
    assert len(loads) >= 3 and all("sc1" in body[i] for i in loads), [body[i] for i in loads]
    assert all(i > returned for i in loads)
    # 4. the pixels (plain stores, possibly to host memory) are released at system scope before the region is counted
    pixels = max(i for i, l in enumerate(body) if l.startswith("global_store_dwordx") and "sc1" not in l)
    release = idx(lambda l: l.startswith("buffer_wbl2") and "sc0" in l and "sc1" in l, pixels)
    region_bump = idx(lambda l: l.startswith("global_atomic_add"), release)
    assert pixels < release < region_bump
    # 5. the counters' re-arming stores are written through as well
    rearm = [l for l in body[loads[-1]:] if l.startswith("global_store_dword ")]
    assert rearm and all("sc1" in l for l in rearm[:1]), rearm


def test_no_scratch_in_the_plain_variants(pool_kernel_asm):
    """The variants BASELINE configs 2 and 3 run keep their path state in registers: no private segment, and the
    register count that SEVEN (five with specular materials) waves per SIMD need — 72: the path state is re-set between
    items, so that it holds no register across the item code (rt_trace_pool_kernel.hip)."""
    text = "\n".join(pool_kernel_asm)
    for variant, max_vgprs in (("Li0ELb0ELb0ELb0E", 72), ("Li1ELb0ELb0ELb0E", 72), ("Li0ELb0ELb1ELb0E", 96), ("Li1ELb0ELb1ELb0E", 96),
                               ("Li2ELb0ELb0ELb0E", 96)):
        m = re.search(r"\.amdhsa_kernel _ZN10rtdev_fast16k_trace_pool_f64I" + variant + r".*?\.end_amdhsa_kernel", text, re.S)
        assert m, variant
        vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(0)).group(1))
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(0)).group(1))
        assert vgprs <= max_vgprs and scratch == 0, (variant, vgprs, scratch)


def test_bvh_walk_reads_its_nodes_from_lds_not_through_flat_addresses(pool_kernel_asm):
    """The BVH variants walk the node array in LDS (two ds_read_b128 per step) or, for large trees, in global memory (two
    global_load_dwordx4) — two instantiations of one routine.  A pointer that may be either (a select between the two
    inside the routine) turns every node read into a flat load: `random` went from 39 to 51 ms that way in round 4.
    Both forms of the step must be there, and no 128-bit flat load (a node read) anywhere in the kernel."""
    text = "\n".join(pool_kernel_asm)
    for variant in ("Li2ELb0ELb0ELb1E", "Li2ELb1ELb1ELb1E"):
        start = text.index("\n_ZN10rtdev_fast16k_trace_pool_f64I" + variant)
        body = text[start:text.index(".Lfunc_end", start)]
        assert "flat_load_dwordx4" not in body, variant
        steps = [m.start() for m in re.finditer(r"v_pk_fma_f32", body)]
        assert len(steps) == 6, (variant, len(steps))                     # three per step, two instantiations
        assert re.search(r"ds_read_b128[^\n]*\n(?:[^\n]*\n){0,12}?[^\n]*v_pk_fma_f32", body), variant
        assert re.search(r"global_load_dwordx4[^\n]*\n(?:[^\n]*\n){0,12}?[^\n]*v_pk_fma_f32", body), variant


def kernel_body(text, variant):
    start = text.index("\n_ZN10rtdev_fast16k_trace_pool_f64I" + variant)
    return text[start:text.index(".Lfunc_end", start)]


def test_rect_test_narrows_exec_itself(pool_kernel_asm):
    """The free-standing rect test of the linear loop is written out with v_cmpx (rt_trace_common.h: rect_closest_update):
    six comparisons that narrow the exec mask themselves, between ONE save and ONE restore of it — the compiler's form
    ANDs four comparison masks in the scalar unit and saves / restores exec twice per rect (8 scalar instructions; the
    scalar unit is what the rects-only variant waits for: C3 72.3 -> 69.5 ms)."""
    text = "\n".join(pool_kernel_asm)
    for variant in ("Li0ELb0ELb0ELb0E", "Li2ELb0ELb0ELb0E"):
        body = kernel_body(text, variant)
        blocks = re.findall(r";;#ASMSTART\n(.*?);;#ASMEND", body, re.S)
        assert "_dpp" not in body and "quad_perm" not in body   # (a DPP op within five instructions of a v_cmpx would need wait states)
        rects = [b for b in blocks if "v_cmpx_ngt_f64" in b]
        assert len(rects) >= 9, (variant, len(rects))     # three planes x (two records of a pair + the odd one out)
        for b in rects:
            ops = [l.split()[0] for l in b.strip().split("\n") if l.strip() and not l.strip().endswith(":")]
            assert ops.count("v_cmpx_ngt_f64") + ops.count("v_cmpx_nlt_f64") == 6, ops
            assert [o for o in ops if o.startswith("s_")] == ["s_mov_b64", "s_cbranch_execz", "s_mov_b64"], ops


def test_the_two_item_variants_add_integers_the_others_doubles(pool_kernel_asm):
    """Per-pixel sums in LDS: ds_add_u64 (fixed point, order-independent) in the variants that keep two items in flight
    (any primitive kind, BVH), ds_add_f64 in the rects-only / spheres-only ones — and never both in one kernel."""
    text = "\n".join(pool_kernel_asm)
    for variant, fixed in (("Li0ELb0ELb0ELb0E", False), ("Li1ELb1ELb1ELb0E", False), ("Li2ELb0ELb0ELb0E", True), ("Li2ELb1ELb1ELb1E", True)):
        body = kernel_body(text, variant)
        assert ("ds_add_u64" in body) == fixed and ("ds_add_f64" in body) == (not fixed), variant
