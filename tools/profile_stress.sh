#!/bin/bash
# Memory-side counters of the 2^20-sphere scene (run through gpurun from the repository root):
# bytes fetched from HBM per launch of the render kernel and the rate that makes, for the reference tree and the device LBVH.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_stress
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--scene stress --spheres 1048576 --nx 3840 --ny 2160 --spp 64 --reps 1"
for v in ref lbvh; do
  extra=""; if [ $v = lbvh ]; then extra="--lbvh"; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v}_trace -- python3 $R/bench_scenes.py $ARGS $extra > $OUT/${v}_trace.log 2>&1 || { tail -5 $OUT/${v}_trace.log; exit 1; }
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/${v}_fetch -- python3 $R/bench_scenes.py $ARGS $extra > $OUT/${v}_fetch.log 2>&1 || { tail -5 $OUT/${v}_fetch.log; exit 1; }
  timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/${v}_l2 -- python3 $R/bench_scenes.py $ARGS $extra > $OUT/${v}_l2.log 2>&1 || { tail -5 $OUT/${v}_l2.log; exit 1; }
done
python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/prof_stress'
print("# 2^20 spheres, 3840x2160 @ 64 spp: production render kernel (the launch with the shortest duration of each run)")
import json
for v in ("ref","lbvh"):
    dur=None
    for p in glob.glob(out+'/%s_trace/*/*_kernel_trace.csv'%v):
        for row in csv.DictReader(open(p)):
            if 'render_' in row['Kernel_Name'] and 'kernel<false' in row['Kernel_Name']:
                d=(int(row['End_Timestamp'])-int(row['Start_Timestamp']))/1e6
                dur=d if dur is None else min(dur,d)
    c=collections.defaultdict(list)
    for p in glob.glob(out+'/%s_fetch/*/*_counter_collection.csv'%v)+glob.glob(out+'/%s_l2/*/*_counter_collection.csv'%v):
        for row in csv.DictReader(open(p)):
            if 'render_' in row['Kernel_Name'] and 'kernel<false' in row['Kernel_Name']:
                c[row['Counter_Name']].append(float(row['Counter_Value']))
    fetch=min(c['FETCH_SIZE'])*1024*2 if c['FETCH_SIZE'] else float('nan')
    if v == "ref":
        json.dump({"nx": 3840, "ny": 2160, "spp": 64, "read_bytes": fetch, "write_bytes": 0, "kernel_ms": dur,
                   "note": "FETCH_SIZE KiB x 1024 x 2 of the production render kernel, reference tree"}, open(out + "/stress_traffic.json", "w"))
    print("%-5s kernel %.3f ms; HBM read %.1f MB per launch (FETCH_SIZE KiB x 1024 x 2) = %.1f GB/s; L2 hits %.3g, misses %.3g, L1->L2 read requests %.3g" % (
        v, dur, fetch/1e6, fetch/(dur*1e-3)/1e9, min(c['TCC_HIT_sum'] or [0]), min(c['TCC_MISS_sum'] or [0]), min(c['TCP_TCC_READ_REQ_sum'] or [0])))
PY
