#!/bin/bash
# usage (GPU box): tools/tt_bounce0.sh <variant|default>[:opt=value,...] ...   duration of the closest-hit launch over the camera rays (kernel trace, one frame at a time)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  v=${spec%%:*}; opts=""; [ "$spec" != "$v" ] && for o in $(echo ${spec#*:} | tr ',' ' '); do opts="$opts --opt $o"; done
  lib=""; [ "$v" != "default" ] && lib=$PWD/build/variants/librrt_$v.so
  tag=$(echo $spec | tr ':=,' '___')
  rm -rf gpurun_out/tb_$tag
  RRT_LIBRARY=$lib timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tb_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1 $opts > gpurun_out/tb_$tag.log 2>&1
  t=$(find gpurun_out/tb_$tag -name "*kernel_trace.csv" | head -1)
  python3 - "$t" "$spec" <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
# the closest-hit launch that follows a camera kernel = bounce 0
out = []; after_rg = False
for r in rows:
    n = r["Kernel_Name"]
    if "k_raygen_aux2_f32" in n: after_rg = True
    elif after_rg and ("k_trace_tiles_f32" in n or "k_trace_pt_f32<false" in n):
        out.append(dur(r)); after_rg = False
print(f"{sys.argv[2]:40s} bounce-0 closest-hit launches (us): " + " ".join(f"{x:.0f}" for x in out))
PY
  rm -rf gpurun_out/tb_$tag
done
