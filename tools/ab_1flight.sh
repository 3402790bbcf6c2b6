#!/bin/bash
# usage (GPU box): tools/ab_1flight.sh <variant|default>[:opt=value[,opt=value]] ...   one frame at a time: per-kernel times are wall times
for spec in "$@"; do
  v=${spec%%:*}; opts=""; [ "$spec" != "$v" ] && for o in $(echo ${spec#*:} | tr ',' ' '); do opts="$opts --opt $o"; done
  lib=""; [ "$v" != "default" ] && lib=$PWD/build/variants/librrt_$v.so
  RRT_LIBRARY=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 --frames-in-flight 1 $opts > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$spec: failed"; tail -3 gpurun_out/ab_tmp.err; continue; }
  python - "$spec" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
k = d["kernel_ms_per_frame"]
print(f"{sys.argv[1]:40s} frame {d['ms_per_step']:7.3f} ms  closest {k['ms_closest']:6.2f}  any {k['ms_any']:6.2f}  raygen {k['ms_raygen']:6.2f}  shade {k['ms_shade']:5.2f}  film {k.get('ms_film', 0):5.2f}")
PY
done
