#!/bin/bash
# A/B of library builds under variants/ on the GPU box (run from the repository root): two interleaved rounds each.  Extra arguments go to bench.py.
for round in 1 2; do
  for f in variants/*.so; do
    RT_AMD_LIB=$PWD/$f timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stats "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$f', d['value'], 'Mpaths/s', d['ms_per_step'], 'ms')
" || exit 1
  done
done
