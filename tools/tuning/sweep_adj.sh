#!/bin/bash
for shape in "128 1000" "256 1000" "512 1000" "1024 1000" "2048 1000" "4096 1000" "64 10000" "256 10000" "1000 10000"; do
  set -- $shape
  SSDE_DRIFT_MIN_TRACKS=32 python tools/bench_colvar.py --tracks $1 --rows $2 --evals 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$1 x $2', d['kernel'], d['engine_kernel'], '%.4f ms/eval kernel %.4f w%d' % (d['ms_per_eval'], d['main_kernel_ms'], d['windows']))"
done
