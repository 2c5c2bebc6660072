#!/bin/bash
# Which level serves the L2 misses of the 2^20-sphere scene's render kernel (run through gpurun from the repository root).
# rocprofv3 on gfx950 lists no Infinity-Cache hit/miss counter (rocprofv3 -L: nothing named MALL; TCC_EA0_RDREQ_DRAM counts
# requests to the DRAM address space, cached in the Infinity Cache or not), so the level is read off the latency of the
# L2's fabric reads: TCC_EA0_RDREQ_LEVEL accumulates the reads outstanding in every L2 cycle, so LEVEL / RDREQ is the mean
# time a fabric read is outstanding, in L2 clocks.  The same two counters on tools/memlevel_calib (random line reads
# from tables of 64 MiB, 192 MiB, 1 GiB and 8 GiB at the render kernel's occupancy; built beforehand with
# hipcc -O3 --offload-arch=gfx950 -o tools/memlevel_calib tools/memlevel_calib.hip) give the figures for reads the
# Infinity Cache serves and for reads HBM serves.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_memlevel
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/stress -- python3 $R/bench_scenes.py --scene stress --spheres 1048576 --nx 3840 --ny 2160 --spp 64 --reps 1 > $OUT/stress.log 2>&1 || { tail -5 $OUT/stress.log; exit 1; }
timeout -s KILL 120 $R/tools/memlevel_calib > $OUT/calib_times.txt 2>&1 || { cat $OUT/calib_times.txt; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/calib -- $R/tools/memlevel_calib > $OUT/calib.log 2>&1 || { tail -5 $OUT/calib.log; exit 1; }
cat $OUT/calib_times.txt | tee $OUT/summary.txt
OUT=$OUT python3 - <<'PY' | tee -a $OUT/summary.txt
import csv, glob, os, collections
out = os.environ['OUT']
def launches(tag, pick):
    d = collections.defaultdict(dict)
    for p in glob.glob(out + '/%s/*/*_counter_collection.csv' % tag):
        for row in csv.DictReader(open(p)):
            if pick(row['Kernel_Name']):
                d[(int(row['Dispatch_Id']), row['Kernel_Name'])][row['Counter_Name']] = float(row['Counter_Value'])
    return sorted(d.items())
def show(what, c):
    rd, lvl, dram, cyc = c['TCC_EA0_RDREQ_sum'], c['TCC_EA0_RDREQ_LEVEL_sum'], c['TCC_EA0_RDREQ_DRAM_sum'], c['GRBM_GUI_ACTIVE'] / 8
    print("%s\n  fabric read requests %.4g (to the DRAM address space %.4g), %.4g shader cycles, %.3f reads per cycle; mean time a read is outstanding: %.0f L2 clocks"
          % (what, rd, dram, cyc, rd / cyc, lvl / rd))
st = launches("stress", lambda k: 'render_' in k and 'kernel<false' in k)
if st:
    show("render kernel, 2^20 spheres 3840x2160 @ 64 (the launch with the fewest fabric reads)", min(st, key=lambda kv: kv[1]['TCC_EA0_RDREQ_sum'])[1])
for (i, name), c in launches("calib", lambda k: 'gather_lines' in k or 'chase_lines' in k):
    show("calibration launch %d %s" % (i, name.split('(')[0]), c)
PY
