#!/bin/bash
# tools/collect_valu.sh TAG  (run ON THE GPU BOX): SQ counters of every kernel family the bench line names, one rocprofv3 --pmc pass
# per workload (a counter pass of its own: no trace domains).  Output: gpurun_out/valu_TAG/<workload>.{csv,json}; tools/summarise_valu.py
# TAG turns them into profiles/valu_per_row.json and profiles/TAG_sq_counters.txt.
TAG=${1:-r04}
ROOT=$PWD
OUT=$ROOT/gpurun_out/valu_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CNT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
one() {   # NAME REGEX CMD...
  local name=$1 regex=$2; shift 2
  ( cd /tmp && rocprofv3 --pmc $CNT --kernel-include-regex "$regex" -d "$OUT/raw_$name" -o c --output-format csv -- "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" ) || echo "[valu] $name: pass exited non-zero"
  find "$OUT/raw_$name" -name "*counter_collection.csv" -exec cp {} "$OUT/$name.csv" \; 2>/dev/null
  rm -rf "$OUT/raw_$name"
  echo "[valu] $name done"
}
for w in headline share8 headline_ou c2 irregular missing missing_one row_varying argos; do
  one $w "iso_|tv_|direct_|dense_" python3 $ROOT/tools/valu_workload.py $w 3
done
one drift "iso_drift" python3 $ROOT/tools/bench_drift.py 10000 10000 9
one few "iso_few|iso_colvar|iso_adj" python3 $ROOT/tools/bench_colvar.py --linear --only lane=track --evals 4
one direct_c3 "direct_fast_kernel" python3 $ROOT/tools/bench_c3.py
one row_varying_h "iso_adj|iso_colvar" python3 $ROOT/tools/bench_colvar.py --with-h --only adjoint --evals 4
find "$OUT" -type f -size +8M -print -delete
du -sh "$OUT"
ls "$OUT"
