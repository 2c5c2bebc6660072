#!/bin/bash
# The large scenes with the world tree rebuilt on the device (run through gpurun from the repository root).
mkdir -p gpurun_out
: > gpurun_out/scenes_lbvh.log
run() { timeout -k 10 300 python bench_scenes.py "$@" 2>/dev/null | grep '^{' >> gpurun_out/scenes_lbvh.log || exit 1; }
run --scene random --nx 1600 --ny 900 --spp 64 --lbvh
run --scene final --nx 1920 --ny 1080 --spp 64 --lbvh
run --scene stress --spheres 1048576 --nx 1920 --ny 1080 --spp 8 --lbvh
python - <<'PY'
import json
for l in open('gpurun_out/scenes_lbvh.log'):
    d=json.loads(l); print(d['scene'], d['tree'], 'lbvh_build_s', d['lbvh_build_s'], 'kernel_ms', d['kernel_ms'], 'Mpaths/s', d['Mpaths_per_s'], 'Mrays/s', d['Mrays_per_s'], 'nodes/ray', d['node_tests_per_ray'])
PY
