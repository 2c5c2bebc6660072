#!/bin/bash
# Sweep of the finished-ray count that ends a traversal phase of render_stack_kernel (run through gpurun from the repository root).
for n in ${SWEEP:-8 16 24 32 40 48 56 64}; do
  echo "RT_READY_LANES=$n"
  RT_READY_LANES=$n bash tools/stack_ab.sh || exit 1
done
