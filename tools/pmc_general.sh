#!/bin/bash
# tools/pmc_general.sh -- SQ issue/stall counters of the general register kernels (run ON THE GPU BOX): where the wave
# cycles of iso_mask_kernel / iso_mask_light_kernel go.  Output: gpurun_out/pmc_general/summary.txt
set -e
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_general
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    --kernel-include-regex "iso_mask" -d "$OUT/raw" -o g --output-format csv -- python3 $ROOT/tools/bench_general.py > "$OUT/run.log" 2>&1 || echo "pmc pass exited non-zero"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[k] += 1
with open(out + "/summary.txt", "w") as fh:
    for k, c in sorted(acc.items()):
        wc = c["SQ_WAVE_CYCLES"] or 1.0
        line = (f"{k}: launches {cnt[k]}, of the wave cycles: parked on s_waitcnt {c['SQ_WAIT_ANY'] / wc:.1%}, issue stall {c['SQ_WAIT_INST_ANY'] / wc:.1%}, "
                f"issuing {c['SQ_ACTIVE_INST_ANY'] / wc:.1%} (VALU {c['SQ_ACTIVE_INST_VALU'] / wc:.1%}); VALU instructions per launch {c['SQ_INSTS_VALU'] / max(cnt[k], 1):.3e}")
        print(line)
        fh.write(line + "\n")
PY
find "$OUT/raw" -name "*.csv" -size +4M -delete
