#!/bin/bash
# Quick PMC passes of the render kernel at 200 spp (run through gpurun from the repository root).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 1 --warmup 0 --spp 200 --no-cpu-baseline --no-stats > $OUT/$name.log 2>&1 || exit 1
}
pass pmc1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pass pmc2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc'
agg=collections.defaultdict(float)
for p in sorted(glob.glob(out+'/pmc*/*/*_counter_collection.csv')):
    for row in csv.DictReader(open(p)):
        if 'render_kernel' in row['Kernel_Name']:
            agg[row['Counter_Name']]+=float(row['Counter_Value'])
rays=522e6  # 1024*1024*200*2.489
it=rays/64
for k,v in sorted(agg.items()): print("%-24s %.4g   per ray-wave %.1f" % (k,v,v/it))
cyc=agg['GRBM_GUI_ACTIVE']/8
print("cycles per SIMD %.4g ; VALU insts per SIMD-cycle %.3f ; lane util %.3f" % (cyc, agg['SQ_INSTS_VALU']/(1024*cyc), agg['SQ_THREAD_CYCLES_VALU']/(agg['SQ_ACTIVE_INST_VALU']*64)))
PY
