#!/usr/bin/env python3
"""Per-dispatch durations of one frame from a rocprofv3 --kernel-trace CSV (bench.py --frames-in-flight 1): the launches of the LAST whole frame, in start order.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pl -- python3 bench.py --no-cpu-baseline --frames-in-flight 1 --steps 3 --warmup 1
    python tools/per_launch.py gpurun_out/pl
"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_raygen_main" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rrtd::", "")
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if dur < 20:
        continue
    print("%9.1f us  +%8.1f us  %-60s grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, dur, name[:60], r.get("Grid_Size", "")))
print("frame: %.2f ms" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e6))
