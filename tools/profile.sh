#!/bin/bash
# Profiling recipe for the GPU box (run through gpurun from the repository root):
#   tools/profile.sh <tag>      writes gpurun_out/prof_<tag>/
# 1) kernel trace + stats of the default benchmark command;
# 2) PMC passes of one launch of the headline workload, one counter set per run;
# 3) a text summary (per-kernel durations, counters per launch, HBM traffic per launch).
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
pass() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stats --no-other-configs > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }
}
pass pmc1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pass pmc2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS
pass pmc3 FETCH_SIZE GRBM_GUI_ACTIVE
pass pmc4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 $R/tools/profile_summary.py $OUT > $OUT/summary.txt || exit 1
cat $OUT/summary.txt
