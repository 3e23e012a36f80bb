#!/usr/bin/env python3
"""EXECUTED-instruction histogram of one variant of k_trace_pool_f64 (run on the GPU box).

    tools/bb_profile.py <workload: c3|c2|c4|boxes|emissive|random> <variant, e.g. Li0ELb0ELb0ELb0E> [out.txt]

Needs racer-tracer_amd/build/libracer_tracer_amd_bb_<variant>.so and build/bb/blocks_<variant>.json (tools/bb_build.sh,
built in the container).  Renders the workload's frame ONCE through the instrumented library (rt_render_frame_device,
what bench.py times), reads the per-block visit counts and multiplies them with the blocks' instructions:

  * totals per issue unit (vector / scalar / scalar memory / LDS / vector memory) and per segment,
  * vector instructions by the classes the PMC counters use, checked against profiles/<round>_<workload>_pmc_summary.json
    when that was collected on the same sources,
  * the opcode histogram (every opcode above 0.2 % of its unit),
  * vector + scalar instructions by SOURCE LINE (top 60) and by source function.

Counts are wave-level issues (SQ_INSTS_* semantics), exact for the launch — the inserted counters change timing, not
control flow: the frame is deterministic and every item runs the same iterations whichever wave takes it.
"""
import collections
import ctypes as C
import importlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

F64_ADD = {"v_add_f64"}
F64_MUL = {"v_mul_f64"}
F64_FMA = {"v_fma_f64", "v_fmac_f64_e32", "v_fmac_f64_e64", "v_fmac_f64"}
F64_TRANS = {"v_rcp_f64_e32", "v_rsq_f64_e32", "v_sqrt_f64_e32", "v_rcp_f64_e64", "v_rsq_f64_e64", "v_sqrt_f64_e64"}


def unit_of(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache", "s_store", "s_atomic")):
        return "smem"
    if op.startswith(("s_branch", "s_cbranch", "s_setpc", "s_swappc", "s_endpgm", "s_call")):
        return "branch"   # SQ_INSTS_BRANCH, not SQ_INSTS_SALU
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio", "s_sethalt", "s_trap", "s_icache", "s_ttrace", "s_inst_prefetch")):
        return "wait/nop"  # internal: no functional unit
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def short(op):
    return re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", op)


def valu_class(op):
    o = short(op)
    if o in ("v_add_f64",):
        return "add_f64"
    if o == "v_mul_f64":
        return "mul_f64"
    if o in ("v_fma_f64", "v_fmac_f64"):
        return "fma_f64"
    if o in ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"):
        return "trans_f64"
    if o == "v_mad_u64_u32" or o.endswith(("_u64", "_i64", "_b64")) and not o.startswith(("v_mov", "v_cmp")):
        return "int64"
    if o.startswith("v_cvt"):
        return "cvt"
    if re.match(r"v_(add|sub|subrev|mul_lo|mul_hi|mad|and|or|xor|not|lshl|lshr|ashr|bfe|bfi|min|max|add3|lshl_add|add_lshl|and_or|or3|xad|alignbit|bitop3|perm|bcnt|ffb|sad)", o) \
            and re.search(r"_(u32|i32|b32|u24|i24|u16|i16)$", o):
        return "int32"
    if re.search(r"_f32$", o) and not o.startswith(("v_cmp", "v_cvt")):
        return "f32"
    return "other"


def issue_cycles(op):
    """SIMD cycles a wave instruction of this opcode holds the vector pipe, as measured on gfx950
    (tools/microbench/valu_cost.hip, profiles/r02_valu_cost.txt): plain 32-bit VOP1/VOP2 2.3; f64 arithmetic, every
    compare, v_cndmask, 64-bit moves and shifts, 32-bit multiplies, v_mad_u64_u32, three-operand integer ops, packed
    f32, lane ops 4.2; f64 rcp / rsq / sqrt 16; f32 transcendentals 8."""
    o = short(op)
    if o in ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"):
        return 16.0
    if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)_f32", o):
        return 8.0
    if o.startswith(("v_cmp", "v_cndmask", "v_pk_", "v_mad_", "v_mul_lo", "v_mul_hi", "v_mbcnt", "v_readlane", "v_writelane", "v_readfirstlane",
                     "v_add3", "v_lshl_add", "v_add_lshl", "v_and_or", "v_lshl_or", "v_or3", "v_xad", "v_bfi", "v_alignbit", "v_perm", "v_med3",
                     "v_min3", "v_max3", "v_div_fixup", "v_ldexp", "v_fma_f64", "v_fmac_f64", "v_mul_u32_u24", "v_mul_hi_u32_u24", "v_mad_u32_u24")):
        return 4.2
    if o.endswith(("_f64", "_b64", "_u64", "_i64")):
        return 4.2
    if o == "v_bitop3_b32" or o.startswith(("v_fma_f32", "v_fmac_f32", "v_min_f32", "v_max_f32", "v_mul_f32", "v_add_f32", "v_sub_f32")):
        return 2.5
    return 2.3


def source_text(cache, where):
    if not where:
        return ""
    f, _, n = where.partition(":")
    if f not in cache:
        cache[f] = []
        for base in ("racer-tracer_amd/csrc", "include"):
            p = os.path.join(ROOT, base, f)
            if os.path.exists(p):
                cache[f] = open(p).read().split("\n")
    lines = cache[f]
    n = int(n)
    return lines[n - 1].strip()[:110] if 0 < n <= len(lines) else ""


def function_of(cache, fcache, where):
    """Name of the function (or lambda-holding function) whose definition precedes the line: a crude scan for
    `name(...) {` at low indentation, good enough to bucket philox / rcp / rect_t / box_slab_t / the kernel body."""
    if not where:
        return "(no line)"
    f, _, n = where.partition(":")
    source_text(cache, where)
    key = (f, int(n))
    if key in fcache:
        return fcache[key]
    lines = cache.get(f, [])
    name = f
    for i in range(min(int(n), len(lines)) - 1, -1, -1):
        m = re.match(r"^(?:template\s*<[^>]*>\s*)?[\w:<>,\s\*&]*?\b(operator\s*[-+*/]|\w+)\s*\([^;{]*(?:\)\s*(?:const\s*)?\{.*)?$", lines[i])
        if m and not lines[i].startswith((" ", "\t", "#", "//")) and m.group(1) not in ("if", "for", "while", "switch", "return", "defined"):
            name = "%s: %s" % (f, m.group(1))
            break
    fcache[key] = name
    return name


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    workload, variant = sys.argv[1], sys.argv[2]
    out = open(sys.argv[3], "w") if len(sys.argv) > 3 else None
    lib_path = os.path.join(ROOT, "racer-tracer_amd", "build", "libracer_tracer_amd_bb_%s.so" % variant)
    blocks = json.load(open(os.path.join(ROOT, "racer-tracer_amd", "build", "bb", "blocks_%s.json" % variant)))
    os.environ["RACER_TRACER_AMD_LIB"] = lib_path
    import torch
    rt = importlib.import_module("racer-tracer_amd")
    host = importlib.import_module("racer-tracer_amd.host")
    bench = importlib.import_module("bench")
    session, name = bench.load_workload(host, workload, 0)
    p = session.params
    if os.environ.get("BB_SPP"):  # a first contact with a freshly instrumented kernel: few samples, under a timeout
        p.samples = int(os.environ["BB_SPP"])
    lib = rt.lib()
    lib.rtdev_bb_counts.restype = C.c_int
    lib.rtdev_bb_counts.argtypes = [C.POINTER(C.c_ulonglong), C.c_int, C.c_int]
    scene = rt.Scene(session)
    frame = torch.zeros((p.height, p.width, 3), dtype=torch.float64, device="cuda")
    lib.rtdev_bb_counts(None, 0, 1)
    scene.render_frame_device(session.camera, p, frame.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = scene.last_stats()
    n = blocks["n_blocks"]
    lanes_at = blocks.get("lane_counters_at")  # builds with lane counters: enabled lanes of every visit, summed, at [lanes_at + id]
    n_read = n if lanes_at is None else lanes_at + n
    raw = (C.c_ulonglong * n_read)()
    if lib.rtdev_bb_counts(raw, n_read, 0) != n_read:
        sys.exit("rtdev_bb_counts failed")
    counts = list(raw[:n])
    lanes = list(raw[lanes_at:lanes_at + n]) if lanes_at is not None else None
    if not any(counts):
        sys.exit("all counters are zero: variant %s did not run for workload %s" % (variant, workload))
    # the frame must be the product's frame (the counters must not have disturbed anything)
    os.environ.pop("RACER_TRACER_AMD_LIB")
    product = rt.load_library(os.path.join(ROOT, "racer-tracer_amd", "lib", "libracer_tracer_amd.so"))
    sc2 = rt.Scene(session, library=product)
    frame2 = torch.zeros_like(frame)
    sc2.render_frame_device(session.camera, p, frame2.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    same = bool(torch.equal(frame, frame2))
    segs = float(st.segments)

    def emit(s=""):
        print(s)
        if out:
            out.write(s + "\n")

    by_unit = collections.Counter()
    by_op = collections.Counter()
    by_class = collections.Counter()
    by_line = collections.Counter()
    by_line_valu = collections.Counter()
    by_func = collections.Counter()
    by_func_valu = collections.Counter()
    cyc_line = collections.Counter()
    cyc_func = collections.Counter()
    cyc_op = collections.Counter()
    lanes_func = collections.Counter()   # enabled lanes summed over the vector instructions issued (SQ_THREAD_CYCLES-like, in lane-issues)
    lanes_line = collections.Counter()
    lanes_total = 0
    cache, fcache = {}, {}
    for b in blocks["blocks"]:
        c = counts[b["id"]]
        if c == 0:
            continue
        lv = lanes[b["id"]] if lanes else 0  # every instruction of a block runs under the mask the block was entered with
        for op, _operands, where in b["insts"]:
            u = unit_of(op)
            by_unit[u] += c
            by_op[(u, short(op))] += c
            if u == "valu":
                by_class[valu_class(op)] += c
                by_line_valu[where] += c
                by_func_valu[function_of(cache, fcache, where)] += c
                cy = c * issue_cycles(op)
                cyc_line[where] += cy
                cyc_func[function_of(cache, fcache, where)] += cy
                cyc_op[short(op)] += cy
                lanes_func[function_of(cache, fcache, where)] += lv
                lanes_line[where] += lv
                lanes_total += lv
            if u != "other":
                by_line[where] += c
                by_func[function_of(cache, fcache, where)] += c
    emit("# executed-instruction histogram: %s, kernel variant <%s>, one launch" % (name, variant))
    emit("# %s; frame %s the product library's; %d of %d basic blocks visited" %
         (blocks["kernel"], "bit-identical to" if same else "DIFFERENT from", sum(1 for c in counts if c), n))
    emit("# segments %.6g, samples %.6g; wave-level instruction issues (SQ_INSTS_* semantics)" % (segs, float(st.samples)))
    emit()
    emit("unit        instructions   per segment")
    for u in ("valu", "salu", "branch", "wait/nop", "smem", "lds", "vmem", "other"):
        emit("%-8s %15.6g   %8.3f" % (u, by_unit[u], by_unit[u] / segs))
    emit()
    pmc = None
    for rnd in ("r04", "r03"):
        path = os.path.join(ROOT, "profiles", "%s_%s_pmc_summary.json" % (rnd, workload))
        if os.path.exists(path):
            d = json.load(open(path))
            k = [x for x in d if "k_trace_pool_f64" in x]
            if k:
                pmc = (os.path.relpath(path, ROOT), d[k[0]], d["_stamp"])
                break
    total_valu = float(by_unit["valu"])
    emit("vector instructions by class (share of the vector instructions)%s" % ("   | PMC counter of %s, ratio counted / PMC" % pmc[0] if pmc else ""))
    names = {"add_f64": "SQ_INSTS_VALU_ADD_F64", "mul_f64": "SQ_INSTS_VALU_MUL_F64", "fma_f64": "SQ_INSTS_VALU_FMA_F64", "trans_f64": "SQ_INSTS_VALU_TRANS_F64",
             "int64": "SQ_INSTS_VALU_INT64", "int32": "SQ_INSTS_VALU_INT32", "cvt": "SQ_INSTS_VALU_CVT", "f32": None, "other": None}
    for cls in ("add_f64", "mul_f64", "fma_f64", "trans_f64", "int64", "int32", "cvt", "f32", "other"):
        line = "  %-10s %14.6g  %6.2f %%" % (cls, by_class[cls], 100.0 * by_class[cls] / total_valu)
        if pmc and names[cls] and names[cls] in pmc[1]:
            m = pmc[1][names[cls]]["mean"]
            line += "   | %14.6g  %.3f" % (m, by_class[cls] / m if m else float("nan"))
        emit(line)
    if pmc:
        emit("  totals against the PMC summary (sources %s the counted build's):" % ("=" if pmc[2].get("source_sha") == importlib.import_module("source_stamp").kernel_source_sha() else "DIFFER from"))
        for u, cn in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("smem", "SQ_INSTS_SMEM"), ("lds", "SQ_INSTS_LDS")):
            if cn in pmc[1]:
                m = pmc[1][cn]["mean"]
                emit("  %-5s counted %14.6g   %s %14.6g   ratio %.4f" % (u, by_unit[u], cn, m, by_unit[u] / m))
    emit()
    for u in ("valu", "salu", "branch", "wait/nop", "lds", "smem", "vmem"):
        tot = float(by_unit[u]) or 1.0
        emit("%s opcodes (share of the unit's instructions, per segment)" % u)
        rest = 0.0
        for (uu, op), c in sorted(by_op.items(), key=lambda kv: -kv[1]):
            if uu != u:
                continue
            if c / tot < 0.002:
                rest += c
                continue
            emit("  %-28s %14.6g  %6.2f %%  %8.4f" % (op, c, 100.0 * c / tot, c / segs))
        if rest:
            emit("  %-28s %14.6g  %6.2f %%" % ("(opcodes below 0.2 %)", rest, 100.0 * rest / tot))
        emit()
    total_cyc = float(sum(cyc_op.values())) or 1.0
    emit("vector ISSUE CYCLES (instructions x the measured cost of their opcode, tools/microbench/valu_cost.hip): %.4g per launch, %.1f per segment"
         % (total_cyc, total_cyc / segs))
    emit("  by opcode (share of the cycles)")
    for op, cy in cyc_op.most_common(25):
        emit("  %-28s %6.2f %%  %8.3f cycles / segment" % (op, 100.0 * cy / total_cyc, cy / segs))
    emit("  by source function")
    for f, cy in cyc_func.most_common(25):
        emit("  %-44s %6.2f %%  %8.3f" % (f, 100.0 * cy / total_cyc, cy / segs))
    emit("  by source line, top 40")
    for where, cy in cyc_line.most_common(40):
        emit("  %-26s %6.2f %%  %8.3f   %s" % (where or "(no line)", 100.0 * cy / total_cyc, cy / segs, source_text(cache, where)))
    emit()
    emit("vector instructions by source function (share of the vector instructions, per segment)")
    for f, c in by_func_valu.most_common(30):
        emit("  %-44s %14.6g  %6.2f %%  %8.4f" % (f, c, 100.0 * c / total_valu, c / segs))
    emit()
    if lanes:
        # DIVERGENCE: enabled lanes of every block visit.  An instruction issued with L of 64 lanes enabled wastes (64 - L) / 64
        # of its issue; "idle share" = the function's / line's part of ALL idle lane-issues of the launch.
        idle_total = 64.0 * total_valu - lanes_total
        emit("lanes enabled per vector instruction: %.2f of 64 over the launch (%.1f %% of the lane-issues idle)"
             % (lanes_total / total_valu, 100.0 * idle_total / (64.0 * total_valu)))
        if pmc and "SQ_THREAD_CYCLES_VALU" in pmc[1] and "SQ_ACTIVE_INST_VALU" in pmc[1]:
            emit("  (PMC summary, cycle-weighted: SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = %.2f)"
                 % (pmc[1]["SQ_THREAD_CYCLES_VALU"]["mean"] / pmc[1]["SQ_ACTIVE_INST_VALU"]["mean"]))
        emit("  by source function: vector instructions, share, lanes per instruction, share of the idle lane-issues")
        for f, c in by_func_valu.most_common(30):
            emit("  %-44s %14.6g  %6.2f %%  %6.2f  %6.2f %%" % (f, c, 100.0 * c / total_valu, lanes_func[f] / c, 100.0 * (64.0 * c - lanes_func[f]) / idle_total))
        emit("  by source line, top 40 by idle lane-issues: share of the instructions, lanes per instruction, share of the idle lane-issues")
        for where, idle in sorted(((w, 64.0 * by_line_valu[w] - lanes_line[w]) for w in by_line_valu), key=lambda kv: -kv[1])[:40]:
            c = by_line_valu[where]
            emit("  %-26s %6.2f %%  %6.2f  %6.2f %%   %s" % (where or "(no line)", 100.0 * c / total_valu, lanes_line[where] / c, 100.0 * idle / idle_total, source_text(cache, where)))
        emit()
    emit("vector instructions by source line, top 60 (share, per segment)")
    for where, c in by_line_valu.most_common(60):
        emit("  %-26s %6.2f %%  %7.4f   %s" % (where or "(no line)", 100.0 * c / total_valu, c / segs, source_text(cache, where)))
    emit()
    emit("blocks by visits, top 40: id, visits, visits per segment, instructions (v/s), lanes per visit, label, first source line")
    order = sorted(blocks["blocks"], key=lambda b: -counts[b["id"]] * len(b["insts"]))[:40]
    for b in order:
        nv = sum(1 for i in b["insts"] if i[0].startswith("v_"))
        ns = sum(1 for i in b["insts"] if i[0].startswith("s_"))
        first = next((i[2] for i in b["insts"] if i[2]), "")
        lpv = "%5.1f" % (lanes[b["id"]] / counts[b["id"]]) if lanes and counts[b["id"]] else "    -"
        emit("  %4d %14.6g %8.4f  %3d/%3d  %s  %-12s %s" % (b["id"], counts[b["id"]], counts[b["id"]] / segs, nv, ns, lpv, b["label"][:12], first))
    if out:  # the raw visit counts, for tools/bb_query.py
        json.dump({"workload": name, "variant": variant, "segments": segs, "samples": float(st.samples), "counts": counts, "lanes": lanes},
                  open(os.path.splitext(sys.argv[3])[0] + "_counts.json", "w"))
    scene.close()
    sc2.close()


if __name__ == "__main__":
    main()
