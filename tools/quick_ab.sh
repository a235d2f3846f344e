#!/bin/bash
# quick A/B of bench.py variants on the GPU box: prints value / ms_per_step / kernel_ms / gpu_ms per variant
set -o pipefail
run() {
  local tag="$1"; shift
  env "$@" python bench.py --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || { echo "$tag FAILED"; tail -5 /tmp/b.err; return 1; }
  python - "$tag" <<'PY'
import json,sys
j=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
w=j['roofline']['whole_evaluation']
print(f"{sys.argv[1]:28s} value {j['value']:.4e} ms/step {j['ms_per_step']:.4f} kernel {j['roofline']['kernel_ms']:.4f} gpu_ms {w['gpu_ms']:.4f} host {w['host_enqueue_ms']:.4f}")
PY
}
