#!/bin/bash
# The other named configurations at their full image sizes (run through gpurun from the repository root); the 16k-spp Cornell
# variant at 256 spp, since its rate does not depend on the sample count.
mkdir -p gpurun_out
: > gpurun_out/scenes_full.log
run() { timeout -k 10 500 python bench_scenes.py --reps 1 "$@" 2>/dev/null | grep '^{' >> gpurun_out/scenes_full.log || exit 1; }
run --scene random --nx 400 --ny 225 --spp 16
run --scene final --nx 1920 --ny 1080 --spp 4096
run --scene stress --spheres 1048576 --nx 3840 --ny 2160 --spp 64
run --scene smoke --nx 7680 --ny 4320 --spp 256
python - <<'PY'
import json
for l in open('gpurun_out/scenes_full.log'):
    d=json.loads(l); print(d['scene'], d['nx'], d['ny'], d['spp'], 'ms', d['ms_per_image'], 'Mpaths/s', d['Mpaths_per_s'], 'Mrays/s', d['Mrays_per_s'], 'roofline', d['roofline'])
PY
