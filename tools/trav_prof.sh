#!/bin/bash
# usage: tools/trav_prof.sh <tag> [handle options]  -> per-benchmark kernel-only durations from rocprofv3 kernel trace
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tp_$tag -- python3 tools/trav_bench.py "$@" > gpurun_out/tp_$tag.log 2>&1
t=$(find gpurun_out/tp_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if ("k_closest<float, false, false>" in r["Kernel_Name"] or "k_trace_pairs_f32<false>" in r["Kernel_Name"] or "k_trace_pt_f32<false>" in r["Kernel_Name"] or "k_trace_pt_f32<false>" in r["Kernel_Name"])]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
# sequence: 6 benches x 23 launches
labels=["primary","secondary","x0.05","x0.25","x1","x4"]
for i,l in enumerate(labels):
    seg=d[i*23+3:(i+1)*23]
    if seg: print(l, "kernel us: median %.1f min %.1f" % (sorted(seg)[len(seg)//2], min(seg)))
print(rows[0]["Kernel_Name"][:40], "LDS", rows[0]["LDS_Block_Size"], "VGPR", rows[0]["VGPR_Count"], "scratch", rows[0]["Scratch_Size"])
PY
