#!/bin/bash
# Vector L1 hit rate of the render kernel on one bench_scenes.py configuration (run through gpurun from the repository root):
#   tools/l1_scene.sh <tag> <bench_scenes.py arguments>
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/l1_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/p -- python3 $R/bench_scenes.py "$@" --reps 1 > $OUT/p.log 2>&1 || { tail -5 $OUT/p.log; exit 1; }
OUT=$OUT python3 - <<'PY'
import csv, glob, os, collections
c=collections.defaultdict(list)
for p in glob.glob(os.environ['OUT']+'/p/*/*_counter_collection.csv'):
    for row in csv.DictReader(open(p)):
        if 'render_kernel<false' in row['Kernel_Name']:
            c[row['Counter_Name']].append(float(row['Counter_Value']))
v={k:min(x) for k,x in c.items()}
print(v)
a=v.get('TCP_TOTAL_CACHE_ACCESSES_sum'); m=v.get('TCP_TCC_READ_REQ_sum')
if a and m: print("L1 accesses %.3g, read requests to L2 %.3g: L1 hit rate %.3f" % (a, m, 1-m/a))
PY
