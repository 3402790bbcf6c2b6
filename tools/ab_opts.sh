#!/bin/bash
# usage (GPU box): tools/ab_opts.sh <variant|default>[:opt=value[,opt=value]] ...   default bench (two frames in flight), handle options per run
for spec in "$@"; do
  v=${spec%%:*}; opts=""; [ "$spec" != "$v" ] && for o in $(echo ${spec#*:} | tr ',' ' '); do opts="$opts --opt $o"; done
  lib=""; [ "$v" != "default" ] && lib=$PWD/build/variants/librrt_$v.so
  RRT_LIBRARY=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 $opts > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$spec: failed"; tail -3 gpurun_out/ab_tmp.err; continue; }
  python - "$spec" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
k = d["kernel_ms_per_frame"]; r = d["roofline"]
print(f"{sys.argv[1]:44s} frame {d['ms_per_step']:7.3f} ms  closest {k['ms_closest']:6.2f} (alone {r['alone']['avg_launch_ms'] * r['launches']:6.2f})  any {k['ms_any']:6.2f}  raygen {k['ms_raygen']:6.2f}  shade {k['ms_shade']:5.2f}")
PY
done
