#!/bin/bash
# Where the waves of render_stack_kernel spend their cycles (diagnostic build variants/timing.so, -DRT_PHASE_TIMING;
# run through gpurun from the repository root).
for sc in "final --nx 1920 --ny 1080 --spp 128" "random --nx 1600 --ny 900 --spp 32" "stress --nx 1920 --ny 1080 --spp 32"; do
  echo "== $sc (RT_READY_LANES=${RT_READY_LANES:-default})"
  RT_AMD_LIB=$PWD/variants/timing.so timeout -k 10 300 python bench_scenes.py --scene $sc --reps 1 2>&1 >/dev/null | grep -E "phase|shading phases|leaf rounds" || exit 1
done
