#!/bin/bash
# tools/collect_profiles.sh TAG -- run ON THE GPU BOX (through gpurun): rocprofv3 kernel statistics of bench.py,
# the two PMC passes for HBM traffic (separate runs, kernel-trace only, as MI355X_MICROARCH.md prescribes), and the
# timings of the other configurations.  Everything lands in gpurun_out/prof_TAG/; tools/summarise_profiles.py
# turns it into the files kept under profiles/.
set -e
TAG=${1:-r01}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- $BENCH > "$OUT/bench_under_rocprof.log" 2>&1
echo "[collect] kernel stats done"
find "$OUT/stats" -name "*kernel_trace*" -delete        # 10^5 records of the synthetic-data generator's kernels
PMCBENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "ssde" -d "$OUT/pmc_fetch" -o fetch --output-format csv -- $PMCBENCH > "$OUT/pmc_fetch.log" 2>&1 || echo "[collect] FETCH_SIZE pass exited non-zero"
echo "[collect] FETCH_SIZE done"
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "ssde" -d "$OUT/pmc_write" -o write --output-format csv -- $PMCBENCH > "$OUT/pmc_write.log" 2>&1 || echo "[collect] WRITE_SIZE pass exited non-zero"
echo "[collect] WRITE_SIZE done"
cd "$ROOT"
python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
echo "[collect] bench line done"
python3 tools/bench_configs.py > "$OUT/other_configs.txt" 2> "$OUT/other_configs.err"
echo "[collect] other configs done"
python3 tools/bench_tv.py > "$OUT/tv_configs.txt" 2> "$OUT/tv_configs.err"
echo "[collect] tv configs done"
find "$OUT" -name "*.csv" -size +8M -delete
du -sh "$OUT"
ls -R "$OUT" | head -50
