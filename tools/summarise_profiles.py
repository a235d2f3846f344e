#!/usr/bin/env python3
"""tools/summarise_profiles.py TAG -- turn gpurun_out/prof_TAG/ (written by tools/collect_profiles.sh on the GPU box)
into the small files kept under profiles/: kernel statistics of this engine's kernels, per-dispatch PMC bytes,
pmc_latest.json (quoted by bench.py as `roofline.traffic_profiled`, with its tag), the bench line and the other configurations' timings."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
rnd, letter = tag[:3], tag[3:].lstrip("_") or "x"
pre = os.path.join(dst, f"{rnd}_{letter}_")

# --- kernel statistics: this engine's kernels only (the synthetic-data generator launches 10^5 torch kernels) ----
rows = list(csv.DictReader(open(os.path.join(src, "stats", "stats_kernel_stats.csv"))))
with open(pre + "kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        if "ssde::" in r["Name"]:
            w.writerow(r)

# --- PMC passes ---------------------------------------------------------------------------------------------------
def pmc(sub, stem):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(src, sub, f"{stem}_counter_collection.csv"))):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return agg

fetch, write = pmc("pmc_fetch", "fetch"), pmc("pmc_write", "write")
with open(pre + "pmc_bytes.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches", "FETCH_SIZE_KiB_avg", "fetch_bytes_corrected_x2", "WRITE_SIZE_KiB_avg", "write_bytes"])
    for k in fetch:
        fa = sum(fetch[k]) / len(fetch[k])
        wa = sum(write.get(k, [0.0])) / max(1, len(write.get(k, [0.0])))
        w.writerow([k, len(fetch[k]), f"{fa:.1f}", f"{2 * 1024 * fa:.0f}", f"{wa:.1f}", f"{1024 * wa:.0f}"])

main = next(k for k in fetch if "iso_shared_kernel" in k or "iso_mask_kernel" in k)
per = {}
for k in fetch:
    if any(t in k for t in ("iso_shared", "iso_mask", "window_check", "reduce_kernel")):
        per[k] = {"fetch_corrected": 2 * 1024 * sum(fetch[k]) / len(fetch[k]),
                  "write": 1024 * sum(write.get(k, [0.0])) / max(1, len(write.get(k, [0.0])))}
main_bytes = per[main]["fetch_corrected"] + per[main]["write"]
line = json.loads(open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1])
pj = {
    "tag": tag,
    "source": f"tools/collect_profiles.sh {tag}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, "
              "--kernel-include-regex ssde) -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline; MI355X",
    "calibration": "tools/microbench_fetch.hip: 1 GiB read once with coalesced 8-B/lane loads -> FETCH_SIZE = 0.5 of the "
                   "bytes, so FETCH_SIZE (KiB) is doubled (as MI355X_MICROARCH.md prescribes); WRITE_SIZE (KiB) taken as is",
    "per_evaluation_bytes": per,
    "main_kernel": main,
    "main_kernel_bytes": main_bytes,
    "hbm_bytes_per_launch": sum(v["fetch_corrected"] + v["write"] for v in per.values()),
    "algorithmic_bytes_per_launch": line["roofline"]["algo_bytes_per_launch"],
    "note": "traffic < algorithmic because on a regular grid the dt channel (8 of the 24 B/row) is never read",
}
json.dump(pj, open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
# the PMC passes of THIS session (same box, same command minus the CPU baseline): recorded next to the line, under a
# name that says where it comes from; bench.py itself never claims a traffic it did not measure
line["roofline"]["traffic_profiled"] = {"main_kernel_bytes": main_bytes, "evaluation_bytes": pj["hbm_bytes_per_launch"],
                                        "source": f"rocprofv3 --pmc passes of the same session ({tag})", "tag": tag}
json.dump(line, open(pre + "bench_line.json", "w"))
shutil.copy(os.path.join(src, "other_configs.txt"), pre + "other_configs.txt")
shutil.copy(os.path.join(src, "tv_configs.txt"), pre + "tv_configs.txt")
# round 3: strong-scaling shares, the drift kernel, BASELINE configs 4 / 5, SQ counters
for name in ("strong_shares.txt", "strong_shares_comm.txt", "drift.txt", "bench_c4.json", "bench_c5.json",
             f"pmc_{tag}_share8.txt", f"pmc_{tag}_drift.txt"):
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copy(f, pre + name.replace(f"pmc_{tag}_", "pmc_"))
try:   # FETCH_SIZE of the drift kernel (same x2 calibration)
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(src, "pmc_drift_fetch", "fetch_counter_collection.csv")))
            if "iso_drift" in r["Kernel_Name"]]
    if vals:
        with open(pre + "pmc_drift_fetch.txt", "w") as f:
            f.write(f"iso_drift_kernel: {len(vals)} dispatches, FETCH_SIZE avg {sum(vals) / len(vals):.1f} KiB -> {2 * 1024 * sum(vals) / len(vals) / 1e9:.3f} GB per launch "
                    f"(x2 calibration of tools/microbench_fetch.hip); tools/bench_drift.py 10000 10000 9 (OU_SSM d = 1, 9 streamed drift columns: 80 B/row required = 8.0 GB)\n")
except Exception as e:  # noqa: BLE001
    print("no drift PMC:", e)
# round 3, second half: the row-varying tau / nu kernel (k_iso_colvar.hip)
for name in ("colvar.txt", f"pmc_{tag}_colvar.txt", f"pmc_{tag}_colvar_h.txt", f"pmc_{tag}_isofull.txt", "quiet.txt", f"pmc_{tag}_quiet.txt"):
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copy(f, pre + name.replace(f"pmc_{tag}_", "pmc_"))
try:
    f = glob.glob(os.path.join(src, "colvar_stats", "**", "*kernel_stats.csv"), recursive=True)
    if f:
        rows = [r for r in csv.DictReader(open(f[0])) if "ssde" in r["Name"]]
        with open(pre + "colvar_kernel_stats.csv", "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            w.writeheader(); w.writerows(rows)
    with open(pre + "pmc_colvar_fetch.txt", "w") as fh:
        for fam in ("iso_adj", "iso_colvar"):
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(src, "pmc_colvar_fetch", "fetch_counter_collection.csv")))
                    if fam in r["Kernel_Name"]]
            if vals:
                fh.write(f"{fam}_kernel: {len(vals)} dispatches, FETCH_SIZE avg {sum(vals) / len(vals):.1f} KiB -> {2 * 1024 * sum(vals) / len(vals) / 1e9:.3f} GB per launch "
                         f"(x2 calibration of tools/microbench_fetch.hip); tools/bench_colvar.py (1e4 CTCRW x 1e3, 18 design columns of which 9 distinct: 88 B/row "
                         f"required = 0.88 GB; the reverse sweep reads every row twice, plus the warm-up and trailing rows of the time windows and its checkpoints)\n")
except Exception as e:  # noqa: BLE001
    print("no colvar profile:", e)
try:
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(src, "pmc_quiet_fetch", "fetch_counter_collection.csv")))
            if "iso_quiet" in r["Kernel_Name"]]
    if vals:
        with open(pre + "pmc_quiet_fetch.txt", "w") as fh:
            fh.write(f"iso_quiet_kernel: {len(vals)} dispatches, FETCH_SIZE avg {sum(vals) / len(vals):.1f} KiB -> {2 * 1024 * sum(vals) / len(vals) / 1e9:.3f} GB per launch "
                     f"(x2 calibration of tools/microbench_fetch.hip); tools/bench_na.py --per-track 1 CTCRW (1e4 x 1e4 rows, 16 B/row required = 1.6 GB, "
                     f"plus the warm-up rows of the time windows)\n")
except Exception as e:  # noqa: BLE001
    print("no quiet-rows PMC:", e)
# round 4: the table drift, C1's latency, the vignette fit
for name in ("drift_table.txt", f"pmc_{tag}_drift_table.txt", "c1_probe.txt", "fit_vignette.txt"):
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copy(f, pre + name.replace(f"pmc_{tag}_", "pmc_"))
print("wrote", pre + "*", "and profiles/pmc_latest.json; main kernel", main, f"{main_bytes / 1e9:.3f} GB per launch")
