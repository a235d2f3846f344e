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
# round 3: the strong-scaling shares of the metric's batch (predicted curve), the drift kernel, BASELINE configs 4 and 5
python3 tools/bench_strong.py > "$OUT/strong_shares.txt" 2> "$OUT/strong_shares.err"
python3 tools/bench_strong.py --comm > "$OUT/strong_shares_comm.txt" 2>> "$OUT/strong_shares.err"
echo "[collect] strong-scaling shares done"
python3 tools/bench_drift.py 10000 10000 9 > "$OUT/drift.txt" 2> "$OUT/drift.err"
python3 tools/bench_drift.py 10000 1000 18 CTCRW 2 >> "$OUT/drift.txt" 2>> "$OUT/drift.err"
echo "[collect] drift kernel done"
python3 bench.py --config c4 --no-cpu-baseline > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
python3 bench.py --config c5 --no-cpu-baseline > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
echo "[collect] c4 / c5 lines done"
bash tools/pmc_kernel.sh "iso_shared" ${TAG}_share8 -- python3 $ROOT/tools/bench_strong.py --ranks 8 --evals 20 > /dev/null 2>&1 || true
bash tools/pmc_kernel.sh "iso_drift" ${TAG}_drift -- python3 $ROOT/tools/bench_drift.py 10000 10000 9 > /dev/null 2>&1 || true
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "iso_drift" -d "$OUT/pmc_drift_fetch" -o fetch --output-format csv -- python3 $ROOT/tools/bench_drift.py 10000 10000 9 > "$OUT/pmc_drift_fetch.log" 2>&1 ) || true
cp gpurun_out/pmc_${TAG}_share8.txt gpurun_out/pmc_${TAG}_drift.txt "$OUT/" 2>/dev/null || true
# round 3, second half: row-varying tau / nu on lane = track lanes (k_iso_colvar.hip) against the lane = direction path
python3 tools/bench_colvar.py > "$OUT/colvar.txt" 2> "$OUT/colvar.err"
python3 tools/bench_colvar.py --linear >> "$OUT/colvar.txt" 2>> "$OUT/colvar.err"
python3 tools/bench_colvar.py --with-h >> "$OUT/colvar.txt" 2>> "$OUT/colvar.err"
python3 tools/bench_colvar.py --with-h --k1 0 --k2 0 --rows 10000 --evals 6 >> "$OUT/colvar.txt" 2>> "$OUT/colvar.err"
python3 tools/bench_colvar.py --with-h --k1 0 --k2 0 --rows 10000 --evals 6 --irregular --only lane=track >> "$OUT/colvar.txt" 2>> "$OUT/colvar.err"
bash tools/pmc_kernel.sh "iso_full" ${TAG}_isofull -- python3 $ROOT/tools/bench_colvar.py --with-h --k1 0 --k2 0 --rows 10000 --evals 4 --only lane=track > /dev/null 2>&1 || true
cp gpurun_out/pmc_${TAG}_isofull.txt "$OUT/" 2>/dev/null || true
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/colvar_stats" -o stats --output-format csv -- python3 $ROOT/tools/bench_colvar.py --only lane=track > "$OUT/colvar_under_rocprof.log" 2>&1 ) || true
find "$OUT/colvar_stats" -name "*kernel_trace*" -delete 2>/dev/null || true
bash tools/pmc_kernel.sh "iso_adj|iso_colvar" ${TAG}_colvar -- python3 $ROOT/tools/bench_colvar.py --only lane=track --evals 5 > /dev/null 2>&1 || true
bash tools/pmc_kernel.sh "iso_adj|iso_colvar" ${TAG}_colvar_h -- python3 $ROOT/tools/bench_colvar.py --with-h --only lane=track --evals 5 > /dev/null 2>&1 || true
cp gpurun_out/pmc_${TAG}_colvar_h.txt "$OUT/" 2>/dev/null || true
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "iso_adj|iso_colvar" -d "$OUT/pmc_colvar_fetch" -o fetch --output-format csv -- python3 $ROOT/tools/bench_colvar.py --only lane=track --evals 5 > "$OUT/pmc_colvar_fetch.log" 2>&1 ) || true
cp gpurun_out/pmc_${TAG}_colvar.txt "$OUT/" 2>/dev/null || true
echo "[collect] colvar kernel done"
# round 3, last part: one missing row in every track -- quiet rows of the general kernel, against SSDE_NO_QUIET=1
python3 tools/bench_na.py --per-track 1 CTCRW OU_SSM BM_SSM > "$OUT/quiet.txt" 2> "$OUT/quiet.err"
SSDE_NO_QUIET=1 python3 tools/bench_na.py --per-track 1 CTCRW OU_SSM BM_SSM >> "$OUT/quiet.txt" 2>> "$OUT/quiet.err"
python3 tools/bench_na.py --per-track 2 CTCRW >> "$OUT/quiet.txt" 2>> "$OUT/quiet.err"
python3 tools/bench_na.py CTCRW >> "$OUT/quiet.txt" 2>> "$OUT/quiet.err"
bash tools/pmc_kernel.sh "iso_quiet" ${TAG}_quiet -- python3 $ROOT/tools/bench_na.py --per-track 1 CTCRW > /dev/null 2>&1 || true
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "iso_quiet" -d "$OUT/pmc_quiet_fetch" -o fetch --output-format csv -- python3 $ROOT/tools/bench_na.py --per-track 1 CTCRW > "$OUT/pmc_quiet_fetch.log" 2>&1 ) || true
cp gpurun_out/pmc_${TAG}_quiet.txt "$OUT/" 2>/dev/null || true
echo "[collect] quiet rows done"
echo "[collect] SQ counters done"
find "$OUT" -name "*.csv" -size +8M -delete
du -sh "$OUT"
ls -R "$OUT" | head -50
