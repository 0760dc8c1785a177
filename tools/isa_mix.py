#!/usr/bin/env python3
"""Instruction mix of the largest basic blocks of one kernel in the gfx950 assembly of a HIP source.
usage: isa_mix.py <source.hip> <kernel-name-substring> [blocks]
Compiles with `hipcc -O3 -S --cuda-device-only --offload-arch=gfx950` into a temporary file (no GPU needed) and prints,
per basic block, the instruction count and the most frequent opcodes — DESIGN.md section 3 quotes the hot block of
msm_accumulate (one XYZZ mixed addition per trip) from this."""
import collections
import os
import re
import subprocess
import sys
import tempfile

src, kern = sys.argv[1], sys.argv[2]
nblocks = int(sys.argv[3]) if len(sys.argv) > 3 else 4
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(root, "include"),
                           "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(kern) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], None
for l in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = [m.group(1), collections.Counter(), 0]
        blocks.append(cur)
        continue
    t = l.strip()
    if not t or t[0] in ";.":
        continue
    if cur is None:
        cur = ["entry", collections.Counter(), 0]
        blocks.append(cur)
    cur[1][t.split()[0]] += 1
    cur[2] += 1
blocks.sort(key=lambda b: -b[2])
for b in blocks[:nblocks]:
    print("%-12s %6d instructions: %s" % (b[0], b[2], ", ".join("%s %d" % kv for kv in b[1].most_common(10))))
