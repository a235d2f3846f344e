#!/usr/bin/env python3
"""tools/isa_loops.py -- static instruction mix of the loops of one kernel in hipcc's -S output.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o k.s smoothsde_amd/csrc/k_iso.hip
    python tools/isa_loops.py k.s 'iso_mask_kernelILi4ELi2ELi13E' [rows_per_iteration]

Prints, for every innermost loop (a backward branch with no other backward branch inside), the number of VALU
instructions by class: fp64 arithmetic, register traffic (v_mov / v_accvgpr_* / v_cndmask), transcendental, other.
Used to see how far a kernel's issue count per row is from its arithmetic (no GPU needed)."""
import re
import sys
from collections import Counter

path, pat = sys.argv[1], sys.argv[2]
rows = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels, instrs = {}, []
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = len(instrs)
        continue
    m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)$", l)
    if m and not m.group(1).startswith("."):
        instrs.append((m.group(1), m.group(2)))
back = []
for i, (op, args) in enumerate(instrs):
    if op.startswith("s_cbranch") or op == "s_branch":
        t = args.split()[0].rstrip(",")
        if t in labels and labels[t] <= i:
            back.append((labels[t], i))
inner = [b for b in back if not any(o != b and b[0] <= o[0] and o[1] <= b[1] for o in back)]


def classify(op):
    if op.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64")):
        return "fp64"
    if op.startswith(("v_mov", "v_accvgpr", "v_cndmask", "v_readfirstlane", "v_readlane", "v_writelane")):
        return "moves"
    if op.startswith(("v_rcp", "v_exp", "v_log", "v_sqrt", "v_rsq", "v_frexp", "v_ldexp", "v_rndne", "v_cvt", "v_div", "v_trig")):
        return "special"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "scalar"


print(f"kernel {pat}: {len(instrs)} instructions, {len(back)} loops, {len(inner)} innermost")
for a, b in sorted(inner, key=lambda ab: ab[0] - ab[1]):
    c = Counter(classify(op) for op, _ in instrs[a:b + 1])
    valu = c["fp64"] + c["moves"] + c["special"] + c["valu_other"]
    if valu < 20:
        continue
    print(f"  loop of {b - a + 1:5d} instr: VALU {valu:5d} (fp64 {c['fp64']}, moves {c['moves']}, special {c['special']}, "
          f"other {c['valu_other']}) vmem {c['vmem']} lds {c['lds']} scalar {c['scalar']}"
          + (f"  -> {valu / rows:.0f} VALU/row" if rows != 1 else ""))
    mv = Counter(op for op, _ in instrs[a:b + 1] if classify(op) == "moves")
    print("      moves:", dict(mv))
