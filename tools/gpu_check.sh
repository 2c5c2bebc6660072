#!/bin/bash
# GPU-box check: parity tests, then a benchmark line (run through gpurun from the repository root).
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/test.log 2>&1
rc=$?
tail -4 gpurun_out/test.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 3 --warmup 1 "$@" > gpurun_out/bench.log 2>&1 || { tail -20 gpurun_out/bench.log; exit 1; }
python - <<'PY'
import json
for l in open('gpurun_out/bench.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print("VALUE %.1f Mpaths/s  ms/step %.2f  kernel_ms %s  frac %s  flops/path %s  cpu %s" % (d['value'], d['ms_per_step'], r.get('kernel_ms'), r.get('frac'), r.get('flops_per_path'), d['cpu_baseline']))
        print("stats kernel ms", (d.get('event_counts_rank0') or {}).get('kernel_ms'))
PY
