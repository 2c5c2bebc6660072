#!/bin/bash
# Builds the library as it was at a git revision into variants/<name>.so, for an A/B on one GPU box:
#   tools/build_variant_at.sh HEAD~1 prev
set -e
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" raytracing_amd/csrc include | tar -x -C "$TMP"
mkdir -p "$ROOT/variants"
cd "$TMP/raytracing_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -shared -o "$ROOT/variants/$NAME.so" rt_device.hip rt_lbvh.hip rt_multi.hip rt_scene.cpp -ldl
rm -rf "$TMP"
echo "built variants/$NAME.so from $REV"
