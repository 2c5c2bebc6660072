#!/bin/bash
# The stack-traversal render kernel on the three hierarchical scenes (run through gpurun from the repository root):
# kernel time, paths/s, rays/s and roofline fraction of each.  RT_AMD_LIB selects another build of the library
# (tools/ab_scenes.sh runs this for every library under variants/); the RT_* tuning variables of rt_device.hip apply.
run() {
  timeout -k 10 300 python bench_scenes.py "$@" --reps 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  %-8s %4dx%-4d@%-4d %9.3f ms %9.1f Mpaths/s %9.1f Mrays/s  frac %.4f' % (d['scene'], d['nx'], d['ny'], d['spp'], d['kernel_ms'], d['Mpaths_per_s'], d['Mrays_per_s'], d['roofline']['frac']))
" || exit 1
}
run --scene final --nx 1920 --ny 1080 --spp 128 || exit 1
run --scene random --nx 1600 --ny 900 --spp 64 || exit 1
run --scene stress --nx 1920 --ny 1080 --spp 8 || exit 1
