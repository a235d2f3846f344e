#!/usr/bin/env python3
"""tools/kernel_regs.py file.hip [-D...] -- compile one kernel file for gfx950 with --save-temps and list every kernel's
VGPR / AGPR / SGPR / scratch / LDS use (what decides waves per SIMD, and whether anything spills)."""
import os
import re
import subprocess
import sys
import tempfile

src = os.path.abspath(sys.argv[1])
flags = sys.argv[2:]
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "--save-temps=obj", "-c", src,
                    "-o", os.path.join(td, "k.o")] + flags, check=True, cwd=td, stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
    s = open(os.path.join(td, asm)).read()
rows = []
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, body = m.group(1), m.group(2)

    def f(key):
        r = re.search(r"\.amdhsa_" + key + r" (\d+)", body)
        return int(r.group(1)) if r else 0
    total, acc = f("next_free_vgpr"), f("accum_offset")
    rows.append((name, acc if acc else total, total - acc if acc else 0, f("next_free_sgpr"), f("private_segment_fixed_size"), f("group_segment_fixed_size")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
for nm, r in zip(names, rows):
    nm = re.sub(r"^void ssde::", "", nm.replace("(anonymous namespace)::", "")).split("(")[0]
    print(f"{nm:60s} vgpr {r[1]:4d} agpr {r[2]:4d} sgpr {r[3]:4d} scratch {r[4]:6d} lds {r[5]:6d}")
