#!/bin/bash
# Throughput of the other named configurations (run through gpurun from the repository root).
mkdir -p gpurun_out
: > gpurun_out/scenes.log
run() { timeout -k 10 300 python bench_scenes.py "$@" 2>/dev/null | grep '^{' >> gpurun_out/scenes.log || exit 1; }
run --scene smoke --nx 1024 --ny 1024 --spp 256
run --scene random --nx 1600 --ny 900 --spp 64
run --scene final --nx 1920 --ny 1080 --spp 64
run --scene stress --spheres 1048576 --nx 1920 --ny 1080 --spp 8
python - <<'PY'
import json
for l in open('gpurun_out/scenes.log'):
    d=json.loads(l); print(d['scene'], d['nx'], d['ny'], d['spp'], 'kernel_ms', d['kernel_ms'], 'Mpaths/s', d['Mpaths_per_s'], 'Mrays/s', d['Mrays_per_s'], 'nodes/ray', d['node_tests_per_ray'], 'build_s', d['build_s'])
PY
