#!/bin/bash
# on the GPU box: every variant library x diag mode -> kernel ms
DG=${DIAGS:-"0 1 4 5"}
for v in "$@"; do
  for dg in $DG; do
    r=$(SSDE_LIB=$PWD/build/variants/libssde_$v.so SSDE_ADJ_DIAG=$dg python tools/bench_colvar.py --only adjoint 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %.4f w%d chk %.1e' % (d['main_kernel_ms'], d['ms_per_eval'], d['windows'], d['window_check']))")
    echo "$v diag=$dg kernel_ms/eval_ms: $r"
  done
done
