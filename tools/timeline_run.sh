#!/bin/bash
# Work-item timelines of the headline scene at the per-rank sample counts of 1-, 4- and 8-way splits (run through gpurun from the
# repository root after `python tools/timeline_build.py`).
mkdir -p gpurun_out
for spp in 1000 250 125; do
  RT_AMD_LIB=$PWD/variants/timeline.so RT_TIMELINE_OUT=/tmp/tl_$spp.bin timeout -k 10 120 python bench.py --steps 1 --warmup 0 --spp $spp --no-cpu-baseline --no-stats > /dev/null 2>&1 || exit 1
  echo "== $spp spp"; python tools/timeline_summary.py /tmp/tl_$spp.bin | head -4
done
