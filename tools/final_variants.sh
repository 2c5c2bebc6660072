#!/bin/bash
# What the marble texture and the media cost on the final scene (run through gpurun from the repository root).
for v in "" nomarble nomedia nomarble,nomedia; do
  RT_FINAL_VARIANT=$v timeout -k 10 200 python bench_scenes.py --scene final --nx 1920 --ny 1080 --spp 256 --reps 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('variant [$v]', d['kernel_ms'], 'ms', d['Mpaths_per_s'], 'Mpaths/s', d['Mrays_per_s'], 'Mrays/s', 'rays/path', d['rays_per_path'], 'box tests/ray', d['node_tests_per_ray'])
" || exit 1
done
