#!/usr/bin/env python3
"""Where do the executions of one opcode come from?  Offline companion of tools/bb_profile.py.

    tools/bb_query.py <counts.json> <opcode regex> [n]     e.g.  gpurun_out/r04/r04_c3_opcode_hist_counts.json 'v_mov_b(32|64)'

Lists the basic blocks (id, visits per segment, label) and source lines that contribute most executions of the matching
opcodes, with the instructions themselves.  Needs racer-tracer_amd/build/bb/blocks_<variant>.json of the same build."""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    c = json.load(open(sys.argv[1]))
    pat = re.compile(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    blocks = json.load(open(os.path.join(ROOT, "racer-tracer_amd", "build", "bb", "blocks_%s.json" % c["variant"])))["blocks"]
    counts, segs = c["counts"], c["segments"]
    by_line = collections.Counter()
    by_block = collections.Counter()
    total = 0
    for b in blocks:
        n = counts[b["id"]]
        for op, operands, where in b["insts"]:
            if pat.search(op):
                by_line[where] += n
                by_block[b["id"]] += n
                total += n
    print("%s: %.6g executions, %.4f per segment" % (sys.argv[2], total, total / segs))
    print("by source line:")
    for where, n in by_line.most_common(top):
        print("  %-30s %6.2f %%  %.4f / segment" % (where or "(no line)", 100.0 * n / total, n / segs))
    print("by block:")
    for bid, n in by_block.most_common(top):
        b = blocks[bid]
        print("  block %d (%s, %s): %.4f visits / segment, %.2f %% of the matches" % (bid, b["label"], b["func"][16:50], counts[bid] / segs, 100.0 * n / total))
        for op, operands, where in b["insts"]:
            if pat.search(op):
                print("        %-22s %-44s %s" % (op, operands[:44], where))


if __name__ == "__main__":
    main()
