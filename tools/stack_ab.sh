#!/bin/bash
# A/B of the stack-traversal render kernels on the hierarchical scenes (run through gpurun from the repository root).
# RT_OLD_STACK_KERNEL=1 selects the kernel that traces a whole wave's rays to the end before shading.
run() {
  timeout -k 10 300 python bench_scenes.py "$@" --reps 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  %-8s %4dx%-4d@%-4d %9.3f ms %9.1f Mpaths/s %9.1f Mrays/s  frac %.4f' % (d['scene'], d['nx'], d['ny'], d['spp'], d['kernel_ms'], d['Mpaths_per_s'], d['Mrays_per_s'], d['roofline']['frac']))
" || exit 1
}
for old in ${VARIANTS:-1 0}; do
  if [ $old = 1 ]; then export RT_OLD_STACK_KERNEL=1; echo "old kernel"; else unset RT_OLD_STACK_KERNEL; echo "new kernel"; fi
  run --scene final --nx 1920 --ny 1080 --spp 128 || exit 1
  run --scene random --nx 1600 --ny 900 --spp 64 || exit 1
  run --scene stress --nx 1920 --ny 1080 --spp 8 || exit 1
done
