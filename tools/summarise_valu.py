#!/usr/bin/env python3
"""tools/summarise_valu.py TAG -- gpurun_out/valu_TAG/ (tools/collect_valu.sh) -> profiles/valu_per_row.json (VALU wave-instructions
per scored row of every kernel family: the table behind `frac_fp64_issue` in bench.py's line) and profiles/TAG_sq_counters.txt
(where each family's wave cycles go)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", f"valu_{tag}")
# rows the dominant launch scores, for the workloads whose script does not print them
KNOWN_ROWS = {"drift": 10_000 * 9_999, "direct_c3": 10_000 * 9_999, "few": 10_000 * 999, "row_varying_h": 10_000 * 999}
FAMILIES = ("iso_shared_kernel", "iso_mask_kernel", "iso_quiet_kernel", "iso_kernel", "iso_drift_kernel", "iso_drift_general_kernel",
            "iso_colvar_kernel", "iso_adj_kernel", "iso_few_kernel", "iso_full_kernel", "dense_kernel", "tv_filter_kernel", "direct_fast_kernel", "direct_kernel")
table, lines = {}, []
for f in sorted(glob.glob(os.path.join(src, "*.csv"))):
    name = os.path.basename(f)[:-4]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("ssde::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            cnt[k] += 1
    rows = KNOWN_ROWS.get(name)
    meta = {}
    try:
        meta = json.loads([l for l in open(os.path.join(src, name + ".json")).read().splitlines() if l.startswith("{")][-1])
        rows = meta.get("rows_in_launch", rows)
    except Exception:  # noqa: BLE001
        pass
    # the dominant kernel of the workload: the one with the most VALU instructions in total
    main = max((k for k in acc if any(k.startswith(fam) for fam in FAMILIES)), key=lambda k: acc[k]["SQ_INSTS_VALU"], default=None)
    if main is None or not rows:
        lines.append(f"{name}: nothing usable")
        continue
    c, n = acc[main], max(cnt[main], 1)
    v_launch = c["SQ_INSTS_VALU"] / n
    wc = c["SQ_WAVE_CYCLES"] or 1.0
    fam = main.split("<")[0]
    uni = fam == "iso_mask_kernel" and main.rstrip(">").endswith("true")
    key = "iso_mask_kernel<uniform grid>" if uni else fam
    ent = {"valu_per_row": v_launch / rows, "valu_per_launch": v_launch, "rows_in_launch": rows, "workload": name, "kernel": main,
           "source": f"profiles/{tag}_sq_counters.txt ({name})"}
    if key not in table or name in ("headline", "missing", "irregular"):
        table[key] = ent
    table[f"{key} [{name}]"] = ent
    lines.append(f"{name}: {main}: launches {n}, waves per launch {c['SQ_WAVES'] / n:.0f}, VALU wave-instructions per launch {v_launch:.4e} = "
                 f"{v_launch / rows:.2f} per scored row ({rows} rows); of the wave cycles: parked on s_waitcnt {c['SQ_WAIT_ANY'] / wc:.1%}, "
                 f"issue stall {c['SQ_WAIT_INST_ANY'] / wc:.1%}, issuing {c['SQ_ACTIVE_INST_ANY'] / wc:.1%} (VALU {c['SQ_ACTIVE_INST_VALU'] / wc:.1%}); "
                 f"fp64-issue roof {1e3 * v_launch * 4 / (1024 * 2.4e9):.4f} ms" + (f", kernel {meta['kernel_ms']:.4f} ms" if meta.get("kernel_ms") else ""))
json.dump({"tag": tag, "unit": "VALU wave-instructions (SQ_INSTS_VALU) per scored row (track-timestep) of the launch", "simds": 1024,
           "clock_hz": 2.4e9, "cycles_per_fp64_valu": 4, "kernels": table}, open(os.path.join(ROOT, "profiles", "valu_per_row.json"), "w"), indent=1)
open(os.path.join(ROOT, "profiles", f"{tag}_sq_counters.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
