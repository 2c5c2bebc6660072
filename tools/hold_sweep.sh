#!/bin/bash
# Sweep of the share of traversing lanes that must hold a leaf before the wave runs the leaf code (run through gpurun).
for n in ${SWEEP:-64 48 32 16}; do
  echo "RT_HOLD_LANES=$n"
  RT_HOLD_LANES=$n bash tools/stack_ab.sh || exit 1
done
