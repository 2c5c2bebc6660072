#!/bin/bash
# Mean occupancy (waves per SIMD) of the render kernel at several sample counts (run through gpurun from the repository root).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/occ
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spp in 250 1000 4000; do
  timeout -s KILL 150 rocprofv3 --pmc MeanOccupancyPerCU --output-format csv -d $OUT/s$spp -- python3 $R/bench.py --steps 1 --warmup 0 --spp $spp --no-cpu-baseline --no-stats > $OUT/s$spp.log 2>&1 || { tail -5 $OUT/s$spp.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/occ'
for p in sorted(glob.glob(out+'/*/*/*_counter_collection.csv')):
    for row in csv.DictReader(open(p)):
        if 'render_kernel' in row['Kernel_Name']:
            print(p.split('/')[-3], row['Counter_Name'], row['Counter_Value'])
PY
