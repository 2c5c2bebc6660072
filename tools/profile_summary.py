#!/usr/bin/env python3
"""Condenses one tools/profile.sh output directory into a text summary:
the bench line of the traced run, rocprofv3's per-kernel statistics, the PMC
counters of one launch of each kernel and the HBM traffic per launch.

FETCH_SIZE and WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950
FETCH_SIZE tallies each 128-byte read request as 64 bytes, so it is doubled;
WRITE_SIZE is taken as is.
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out = sys.argv[1]
    print("# profile summary of", os.path.basename(out))
    for line in open(os.path.join(out, "trace.log")):
        if line.startswith("{"):
            d = json.loads(line)
            r = d.get("roofline") or {}
            print("bench line under rocprofv3 --kernel-trace --stats: value %.1f %s, ms_per_step %.3f, kernel_ms (HIP events) %s, "
                  "roofline frac %s" % (d["value"], d["unit"], d["ms_per_step"], r.get("kernel_ms"), r.get("frac")))
    print("\n## rocprofv3 --kernel-trace --stats: kernels (python3 bench.py, default flags)")
    for p in glob.glob(os.path.join(out, "trace", "*", "*_kernel_stats.csv")):
        rows = list(csv.DictReader(open(p)))
        print("%-110s %6s %12s %12s %12s %7s" % ("kernel", "calls", "avg_ms", "min_ms", "max_ms", "pct"))
        for row in rows:
            name = row["Name"]
            if len(name) > 108:
                name = name[:105] + "..."
            print("%-110s %6s %12.4f %12.4f %12.4f %7.2f" % (
                name, row["Calls"], float(row["AverageNs"]) / 1e6, float(row["MinNs"]) / 1e6, float(row["MaxNs"]) / 1e6,
                float(row["Percentage"])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for p in sorted(glob.glob(os.path.join(out, "pmc*", "*", "*_counter_collection.csv"))):
        tag = p.split(os.sep)[-3]
        for row in csv.DictReader(open(p)):
            k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            k = k.split("(")[0]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[(k, tag)].add(row["Dispatch_Id"])
    print("\n## PMC, one launch of the headline workload (bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stats --no-other-configs)")
    traffic = {}
    for k, counters in agg.items():
        if "render_kernel" not in k and "render_stack_kernel" not in k and "reduce_slabs" not in k:
            continue
        n = max(len(v) for (kk, _), v in launches.items() if kk == k)
        print("\n%s   (%d launch%s)" % (k, n, "" if n == 1 else "es"))
        for c, v in sorted(counters.items()):
            print("  %-24s %.6g" % (c, v / n))
        fetch = counters.get("FETCH_SIZE", 0.0) / n * 1024.0 * 2.0
        write = counters.get("WRITE_SIZE", 0.0) / n * 1024.0
        print("  HBM bytes per launch: read %.4g (FETCH_SIZE KiB x 1024 x 2), written %.4g (WRITE_SIZE KiB x 1024)" % (fetch, write))
        traffic[k] = {"read_bytes": fetch, "write_bytes": write}
        if "SQ_INSTS_VALU" in counters and "GRBM_GUI_ACTIVE" in counters:
            cyc = counters["GRBM_GUI_ACTIVE"] / n / 8.0  # summed over the 8 XCDs
            print("  VALU instructions per SIMD-cycle: %.3f of 1 (1024 SIMDs, %.4g shader cycles)" % (
                counters["SQ_INSTS_VALU"] / n / (1024.0 * cyc), cyc))
        if "SQ_THREAD_CYCLES_VALU" in counters and "SQ_ACTIVE_INST_VALU" in counters:
            print("  VALU lane utilisation: %.3f" % (counters["SQ_THREAD_CYCLES_VALU"] / (counters["SQ_ACTIVE_INST_VALU"] * 64.0)))
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
