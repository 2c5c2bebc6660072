#!/bin/bash
# A/B of library builds under variants/ on the hierarchical scenes (run through gpurun from the repository root).
for f in variants/*.so; do
  echo "$f"
  RT_AMD_LIB=$PWD/$f bash tools/stack_ab.sh || exit 1
done
