#!/bin/bash
# tools/pmc_kernel.sh REGEX OUTNAME -- CMD...   (run ON THE GPU BOX): SQ issue / stall counters of the kernels whose name
# matches REGEX while CMD runs -- where their wave cycles go and how many VALU instructions they issue.  Counters are
# collected in a pass of their own (no trace domains), as MI355X_MICROARCH.md prescribes.  Output: gpurun_out/pmc_OUTNAME.txt
set -e
REGEX=$1; NAME=$2; shift 3
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_$NAME
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES \
    --kernel-include-regex "$REGEX" -d "$OUT/raw" -o g --output-format csv -- "$@" > "$OUT/run.log" 2>&1 || echo "pmc pass exited non-zero"
cd "$ROOT"
python3 - "$OUT" "$NAME" <<'PY'
import csv, glob, sys, collections
out, name = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[k] += 1
with open(out + "/../pmc_" + name + ".txt", "w") as fh:
    for k, c in sorted(acc.items()):
        wc = c["SQ_WAVE_CYCLES"] or 1.0
        n = max(cnt[k], 1)
        line = (f"{k}: launches {cnt[k]}, waves per launch {c['SQ_WAVES'] / n:.0f}, wave cycles per launch {wc / n:.3e}; of the wave cycles: parked on s_waitcnt "
                f"{c['SQ_WAIT_ANY'] / wc:.1%}, issue stall {c['SQ_WAIT_INST_ANY'] / wc:.1%}, issuing {c['SQ_ACTIVE_INST_ANY'] / wc:.1%} "
                f"(VALU {c['SQ_ACTIVE_INST_VALU'] / wc:.1%}); VALU instructions per launch {c['SQ_INSTS_VALU'] / n:.3e}")
        print(line)
        fh.write(line + "\n")
PY
rm -rf "$OUT/raw"
