#!/bin/bash
# VALU instruction mix of the render kernel at 200 spp (run through gpurun from the repository root).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcmix
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 1 --warmup 0 --spp 200 --no-cpu-baseline --no-stats > $OUT/$name.log 2>&1 || exit 1
}
pass m1 SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
pass m2 SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU2 SQ_INST_LEVEL_SMEM SQ_INSTS_VSKIPPED SQ_IFETCH
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmcmix'
agg=collections.defaultdict(float)
for p in sorted(glob.glob(out+'/m*/*/*_counter_collection.csv')):
    for row in csv.DictReader(open(p)):
        if 'render_kernel' in row['Kernel_Name']:
            agg[row['Counter_Name']]+=float(row['Counter_Value'])
it=522e6/64
for k,v in sorted(agg.items()): print("%-24s %.4g   per ray-wave %.1f" % (k,v,v/it))
PY
