#!/usr/bin/env python3
"""Basic-block execution counters for ONE variant of k_trace_pool_f64, inserted into the compiler's own assembly.

    tools/bb_instrument.py <dev.s> <dev_with_line_tables.s | -> <variant, e.g. Li0ELb0ELb0ELb0E> <out.s> <blocks.json>

What for: the PMC counters say how many vector / scalar / LDS instructions a launch executes and in which arithmetic
class, not WHICH instructions.  This tool answers that exactly.  It takes the device assembly hipcc produces for
csrc/rt_trace_pool_kernel.hip (product flags, `--cuda-device-only -S`), cuts the chosen kernel and the out-of-line
device functions it calls (deliver_item, sphere_uv) into basic blocks, and puts five instructions in front of every
block that add 1 to the block's own counter — lane (id % 64) of VGPR (first free VGPR + id / 64), a register the
kernel does not use:

    s_mov_b64 s[100:101], exec ; s_mov_b32 exec_lo, <bit> ; s_mov_b32 exec_hi, <bit> ; v_add_u32 vC, 1, vC ; s_mov_b64 exec, s[100:101]

(no SCC, VCC or M0 is touched; s100/s101 lie above the 100 SGPRs the kernels use).  A second set of counters takes the
number of ENABLED LANES of every visit (popcount of the exec mask, formed without the scalar ALU: v_mbcnt + the lane's own
bit in lane 63, handed to the block's counter lane through ds_bpermute): visits x instructions are what the wave issues,
lanes / visits is how much of each issue was useful.  The counters are zeroed at kernel
entry and added to the device array `rt_bb_counts` (64-bit atomics) before every s_endpgm.  The code the compiler
generated is otherwise untouched, so   count(block) x instructions(block)   summed over the blocks is what the
product kernel issues, instruction by instruction: tools/bb_profile.py prints it by opcode and checks the class totals
against the PMC counters of the product build.  A count is per WAVE visit of the block (like SQ_INSTS_*), whatever the
exec mask, including visits with no lane enabled.

blocks.json: [{id, func, label, insts: [[opcode, operands, file:line]]}] — source lines come from the second listing
(the same compile with -gline-tables-only), transferred instruction by instruction where the two listings agree
(line tables move a handful of instructions in some variants; those stay without a line).

Only the instrumented variant may run in a process that loaded the instrumented library: the shared callees count
into VGPRs the other kernels do not reserve.
"""
import difflib
import json
import re
import sys

INSTR = re.compile(r"^\t([a-z][a-z0-9_]*)\b\s*(.*)$")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FUNC = re.compile(r"^(_Z\w+):")
TERMINATORS = ("s_branch", "s_cbranch_", "s_endpgm", "s_setpc_b64")
COUNTS_SYMBOL = "rt_bb_counts"
MAX_BLOCKS = 4096   # rt_bb_counts holds RT_BB_MAX = 8192 entries: visits at [id], enabled lanes at [LANES_AT + id]
LANES_AT = 4096


def functions(lines):
    """[(name, first line after the label, line of .Lfunc_end)]"""
    out = []
    i = 0
    while i < len(lines):
        m = FUNC.match(lines[i])
        if m:
            j = i + 1
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            out.append((m.group(1), i + 1, j))
            i = j
        i += 1
    return out


def instruction_of(line):
    m = INSTR.match(line)
    if not m:
        return None
    text = m.group(2).split(";")[0].strip()
    return m.group(1), text


def line_map(lines_g, name):
    """[(opcode, operands, 'file:line')] of function `name` in the listing with line tables."""
    files = {}
    for l in lines_g:
        m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    fs = [f for f in functions(lines_g) if f[0] == name]
    if not fs:
        return []
    _, a, b = fs[0]
    out, cur = [], ""
    for l in lines_g[a:b]:
        m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = "%s:%s" % (files.get(int(m.group(1)), m.group(1)), m.group(2))
            continue
        ins = instruction_of(l)
        if ins:
            out.append((ins[0], ins[1], cur))
    return out


def main():
    if len(sys.argv) != 6:
        sys.exit(__doc__)
    src, src_g, variant, out_s, out_json = sys.argv[1:]
    lines = open(src).read().split("\n")
    lines_g = open(src_g).read().split("\n") if src_g != "-" else []
    funcs = functions(lines)
    kernel = [f for f in funcs if "rtdev_fast" in f[0] and "k_trace_pool_f64I" + variant in f[0]]
    if len(kernel) != 1:
        sys.exit("variant %r matches %d kernels" % (variant, len(kernel)))
    kernel = kernel[0]
    callees = [f for f in funcs if "rtdev_fast" in f[0] and "k_trace_pool_f64" not in f[0] and "k_resolve" not in f[0] and "k_post" not in f[0]]
    # VGPRs of the kernel (its descriptor's literal) and of everything it may call
    desc_at = next(i for i, l in enumerate(lines) if l.strip() == ".amdhsa_kernel " + kernel[0])
    desc_end = next(i for i in range(desc_at, len(lines)) if ".end_amdhsa_kernel" in lines[i])

    def desc_value(key):
        for i in range(desc_at, desc_end):
            m = re.match(r"\s+\." + key + r"\s+(\d+)", lines[i])
            if m:
                return i, int(m.group(1))
        sys.exit("no literal .%s in the descriptor of %s" % (key, kernel[0]))

    vg_line, n_vgpr = desc_value("amdhsa_next_free_vgpr")
    sg_line, n_sgpr = desc_value("amdhsa_next_free_sgpr")
    acc_line, _ = desc_value("amdhsa_accum_offset")
    if n_sgpr > 100:
        sys.exit("the kernel uses %d SGPRs: s100/s101 are not free" % n_sgpr)

    # ---- basic blocks
    blocks = []          # {id, func, label, insts}
    insert_before = {}   # line index -> block id  (the block's first instruction)
    for name, a, b in [kernel] + callees:
        lm = line_map(lines_g, name) if lines_g else []
        own = [(i, instruction_of(lines[i])) for i in range(a, b) if instruction_of(lines[i])]
        where = {}
        if lm:  # transfer the source lines where the two listings agree
            sm = difflib.SequenceMatcher(a=[x[1] for x in own], b=[(o, t) for o, t, _ in lm], autojunk=False)
            for blk in sm.get_matching_blocks():
                for k in range(blk.size):
                    where[own[blk.a + k][0]] = lm[blk.b + k][2]
        cur = None
        label = "entry"
        for i in range(a, b):
            m = LABEL.match(lines[i])
            if m:
                cur = None
                label = m.group(1)
                continue
            ins = instruction_of(lines[i])
            if not ins:
                continue
            if cur is None:
                cur = {"id": len(blocks), "func": name, "label": label, "insts": []}
                blocks.append(cur)
                insert_before[i] = cur["id"]
                label = ""
            cur["insts"].append([ins[0], ins[1], where.get(i, "")])
            if ins[0].startswith(TERMINATORS):
                cur = None
    if len(blocks) > MAX_BLOCKS:
        sys.exit("%d blocks: raise MAX_BLOCKS / LANES_AT (and RT_BB_MAX in rt_trace_pool_kernel.hip)" % len(blocks))
    n_counters = (len(blocks) + 63) // 64
    c0 = n_vgpr                       # visit counters c0 .. c0 + n_counters - 1
    l0 = c0 + n_counters              # lane counters l0 .. l0 + n_counters - 1 (sum of enabled lanes over the visits)
    t_cnt = l0 + n_counters           # scratch: popcount of exec (valid in lane 63), then broadcast
    t_bit = t_cnt + 1                 # scratch: the lane's own exec bit
    t_adr = t_cnt + 2                 # constant 252: byte address of lane 63 for ds_bpermute (set at kernel entry)
    t_off = t_cnt + 3                 # scratch: byte offset of this lane's counter (flush)
    t_pair = (t_off + 2) & ~1         # scratch pair (even-aligned) for the 64-bit atomic's data
    new_vgpr = (t_pair + 2 + 7) & ~7
    if new_vgpr > 512:
        sys.exit("not enough VGPRs")

    def bump(bid):
        lane = bid % 64
        lo, hi = (1 << lane) & 0xffffffff, (1 << lane) >> 32
        v, vl = c0 + bid // 64, l0 + bid // 64
        return ["\ts_mov_b64 s[100:101], exec\t; bb %d" % bid,
                "\ts_mov_b64 exec, -1",
                "\tv_mbcnt_lo_u32_b32 v%d, s100, 0" % t_cnt,
                "\tv_mbcnt_hi_u32_b32 v%d, s101, v%d" % (t_cnt, t_cnt),          # set bits of the saved mask below this lane
                "\tv_cndmask_b32_e64 v%d, 0, 1, s[100:101]" % t_bit,            # this lane's own bit
                "\tv_add_u32_e32 v%d, v%d, v%d" % (t_cnt, t_cnt, t_bit),          # lane 63: popcount(exec)
                "\tds_bpermute_b32 v%d, v%d, v%d" % (t_cnt, t_adr, t_cnt),        # ... to every lane
                "\ts_waitcnt lgkmcnt(0)",
                "\ts_mov_b32 exec_lo, 0x%x" % lo, "\ts_mov_b32 exec_hi, 0x%x" % hi,
                "\tv_add_u32_e32 v%d, 1, v%d" % (v, v),
                "\tv_add_u32_e32 v%d, v%d, v%d" % (vl, t_cnt, vl),
                "\ts_mov_b64 exec, s[100:101]"]

    flush = ["\ts_mov_b64 exec, -1\t; bb flush",
             "\ts_getpc_b64 s[100:101]",
             "\ts_add_u32 s100, s100, %s@rel32@lo+4" % COUNTS_SYMBOL,
             "\ts_addc_u32 s101, s101, %s@rel32@hi+12" % COUNTS_SYMBOL,
             "\tv_mbcnt_lo_u32_b32 v%d, -1, 0" % t_off,
             "\tv_mbcnt_hi_u32_b32 v%d, -1, v%d" % (t_off, t_off),
             "\tv_lshlrev_b32_e32 v%d, 3, v%d" % (t_off, t_off),
             "\tv_mov_b32_e32 v%d, 0" % (t_pair + 1)]
    # lane l of counter register j holds block 64 j + l: entry (base + 64 j + l) of rt_bb_counts, 512 bytes per register
    for base, first in ((0, c0), (LANES_AT, l0)):
        for j in range(n_counters):
            flush += ["\tv_mov_b32_e32 v%d, v%d" % (t_pair, first + j),
                      "\tv_add_u32_e32 v%d, 0x%x, v%d" % (t_cnt, base * 8 + j * 512, t_off),
                      "\tglobal_atomic_add_x2 v%d, v[%d:%d], s[100:101]" % (t_cnt, t_pair, t_pair + 1)]
    flush += ["\ts_waitcnt vmcnt(0)"]

    out = []
    first_kernel_inst = next(i for i in range(kernel[1], kernel[2]) if instruction_of(lines[i]))
    for i, l in enumerate(lines):
        if i == first_kernel_inst:
            out += ["\tv_mov_b32_e32 v%d, 0\t; bb counters" % (c0 + j) for j in range(2 * n_counters)]
            out += ["\tv_mov_b32_e32 v%d, 0xfc\t; lane 63 for ds_bpermute" % t_adr]
        if i in insert_before:
            out += bump(insert_before[i])
        if kernel[1] <= i < kernel[2] and instruction_of(l) and instruction_of(l)[0] == "s_endpgm":
            out += flush
        if i == vg_line:
            l = re.sub(r"\d+", str(new_vgpr), l)
        if i == sg_line:
            l = re.sub(r"\d+", "102", l)
        if i == acc_line:
            l = re.sub(r"\d+", str((new_vgpr + 3) & ~3), l)
        out.append(l)
    text = "\n".join(out)
    # metadata: the kernel's register counts (informational for the runtime, kept consistent)
    meta = re.search(r"(\.name:\s+" + re.escape(kernel[0]) + r"\n.*?\.vgpr_count:\s+)(\d+)", text, re.S)
    if meta:
        text = text[:meta.start(2)] + str(new_vgpr) + text[meta.end(2):]
    open(out_s, "w").write(text)
    json.dump({"kernel": kernel[0], "variant": variant, "n_blocks": len(blocks), "counter_vgprs": [c0, c0 + n_counters - 1], "lane_counters_at": LANES_AT,
               "vgprs_before": n_vgpr, "vgprs_after": new_vgpr, "blocks": blocks}, open(out_json, "w"))
    n_inst = sum(len(b["insts"]) for b in blocks)
    with_line = sum(1 for b in blocks for i in b["insts"] if i[2])
    print("%s: %d blocks, %d instructions (%d with a source line), counters in v%d..v%d, %d -> %d VGPRs"
          % (variant, len(blocks), n_inst, with_line, c0, c0 + n_counters - 1, n_vgpr, new_vgpr))


if __name__ == "__main__":
    main()
