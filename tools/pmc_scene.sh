#!/bin/bash
# Shader-side counters of one bench_scenes.py configuration (run through gpurun from the repository root):
#   tools/pmc_scene.sh <tag> <bench_scenes.py arguments>
# Prints, for the production render kernel: VALU instructions per SIMD-cycle, lane utilisation, and how the waves' time splits
# into issuing, waiting on a counter and waiting to issue.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench_scenes.py $ARGS --reps 1 > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }
}
ARGS="$*"
pass p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pass p2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
OUT=$OUT python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob, os, collections
out=os.environ['OUT']
c=collections.defaultdict(list)
for p in glob.glob(out+'/p*/*/*_counter_collection.csv'):
    for row in csv.DictReader(open(p)):
        if 'render_' in row['Kernel_Name'] and 'kernel<false' in row['Kernel_Name']:
            c[row['Counter_Name']].append(float(row['Counter_Value']))
v={k:min(x) for k,x in c.items()}
for k in sorted(v): print("  %-24s %.5g" % (k, v[k]))
cyc=v['GRBM_GUI_ACTIVE']/8
print("VALU instructions per SIMD-cycle: %.3f of 0.5" % (v['SQ_INSTS_VALU']/(1024*cyc)))
alli = v['SQ_INSTS_VALU'] + v['SQ_INSTS_SALU'] + v['SQ_INSTS_SMEM'] + v['SQ_INSTS_LDS'] + v['SQ_INSTS_VMEM_RD']
print("vector + scalar + memory instructions per SIMD-cycle: %.3f (the flat kernel saturates at 0.5)" % (alli/(1024*cyc)))
print("lane utilisation: %.3f" % (v['SQ_THREAD_CYCLES_VALU']/(v['SQ_ACTIVE_INST_VALU']*4*64/4)/1.0 if False else v['SQ_THREAD_CYCLES_VALU']/(v['SQ_INSTS_VALU']*64)))
w=v['SQ_WAVE_CYCLES']
print("wave time: issuing %.3f, parked on a counter %.3f, waiting to issue %.3f" % (v['SQ_ACTIVE_INST_ANY']/w, v['SQ_WAIT_ANY']/w, v['SQ_WAIT_INST_ANY']/w))
print("waves per SIMD (average): %.2f" % (w*4/(1024*cyc)))
PY
