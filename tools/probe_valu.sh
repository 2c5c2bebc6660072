#!/bin/bash
# What bounds the headline kernel: builds of the library with 50 more instructions of one kind in every trip of the path
# loop (-DRT_PROBE=k of rt_device.hip), timed against the normal build.  Step 1, here (no GPU needed):
#   tools/probe_valu.sh build      -> variants/p0_normal.so, p1_s_nop.so, p2_v_xor.so, p3_v_cmp.so, p4_v_cndmask.so, p5_v_med3.so, p6_s_and.so
# step 2, through gpurun from the repository root:
#   tools/probe_valu.sh run        -> ms per image of each, and the waves-per-CU sweep of the normal build
set -u
names=(normal s_nop v_xor v_cmp v_cndmask v_med3 s_and)
if [ "${1:-}" = build ]; then
  mkdir -p variants
  cd raytracing_amd/csrc || exit 1
  for k in 0 1 2 3 4 5 6; do
    def=""; [ $k -gt 0 ] && def="-DRT_PROBE=$k"
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $def -shared \
      -o ../../variants/p${k}_${names[$k]}.so rt_device.hip rt_lbvh.hip rt_multi.hip rt_scene.cpp -ldl &
    [ $((k % 4)) -eq 3 ] && wait
  done
  wait
  exit 0
fi
mkdir -p gpurun_out
{
  bash tools/ab.sh --no-other-configs || exit 1
  echo "waves per CU (RT_WAVES_PER_CU) of the normal build:"
  for w in 4 8 12 16 20 24 28 32; do
    RT_AMD_LIB=$PWD/variants/p0_normal.so RT_WAVES_PER_CU=$w timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stats --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  $w waves/CU', d['ms_per_step'], 'ms')
" || exit 1
  done
} 2>&1 | tee gpurun_out/probe_valu.txt
