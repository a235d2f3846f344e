#!/bin/bash
# tools/collect_round4.sh TAG -- run ON THE GPU BOX after tools/collect_profiles.sh TAG (a call of its own: gpurun's limit is 20 minutes):
# the table drift against the streamed one, C1's latency probe, the vignette fit.  Lands in gpurun_out/prof_TAG/ like the rest.
TAG=${1:-r04}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# round 4: the drift's block as a table evaluated by the lanes (k_iso_drift_pp.hip) against the streamed columns; C1's latency
for m in "OU_SSM 1" "BM_SSM 1" "OU_SSM 2"; do
  set -- $m
  python3 tools/bench_drift.py 10000 10000 9 $1 $2 0 2>> "$OUT/drift_table.err" | head -1 >> "$OUT/drift_table.txt"
  python3 tools/bench_drift.py 10000 10000 9 $1 $2 0 table 2>> "$OUT/drift_table.err" | head -1 >> "$OUT/drift_table.txt"
done
SSDE_DRIFT_PP_ALL=1 python3 tools/bench_drift.py 10000 10000 9 CTCRW 1 0 table 2>> "$OUT/drift_table.err" | head -1 >> "$OUT/drift_table.txt"
python3 tools/bench_drift.py 10000 10000 9 CTCRW 1 0 2>> "$OUT/drift_table.err" | head -1 >> "$OUT/drift_table.txt"
SSDE_DRIFT_PP_ALL=1 python3 tools/bench_drift.py 10000 10000 9 OU_SSM 1 0.02 table 2>> "$OUT/drift_table.err" | head -1 >> "$OUT/drift_table.txt"
python3 tools/bench_drift.py 10000 10000 9 OU_SSM 1 0.02 2>> "$OUT/drift_table.err" | head -1 >> "$OUT/drift_table.txt"
bash tools/pmc_kernel.sh "iso_drift" ${TAG}_drift_table -- python3 $ROOT/tools/bench_drift.py 10000 10000 9 OU_SSM 1 0 table > /dev/null 2>&1 || true
cp gpurun_out/pmc_${TAG}_drift_table.txt "$OUT/" 2>/dev/null || true
echo "[collect] table drift done"
python3 tools/probe_c1.py > "$OUT/c1_probe.txt" 2> "$OUT/c1_probe.err"
SSDE_TV_NO_LEAN=1 python3 tools/probe_c1.py >> "$OUT/c1_probe.txt" 2>> "$OUT/c1_probe.err"
SSDE_TV_MINLEN=32 python3 tools/probe_c1.py >> "$OUT/c1_probe.txt" 2>> "$OUT/c1_probe.err"
python3 tools/probe_c1.py --with-h >> "$OUT/c1_probe.txt" 2>> "$OUT/c1_probe.err"
echo "[collect] C1 probe done"
python3 tools/fit_vignette.py > "$OUT/fit_vignette.txt" 2> "$OUT/fit_vignette.err" || true
echo "[collect] vignette fit done"
du -sh "$OUT"
