#!/bin/bash
# usage (GPU box): tools/variant_bench.sh <name> [<name> ...]   ("default" = rs_ray_toy_amd/csrc/librrt.so)
# one short bench.py run per kernel-tuning variant built by tools/build_variant.sh; prints frame time and per-kernel times
for v in "$@"; do
  lib=""; [ "$v" != "default" ] && lib=$PWD/build/variants/librrt_$v.so
  RRT_LIBRARY=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/vb_$v.json 2> gpurun_out/vb_$v.err || { echo "$v: failed"; tail -3 gpurun_out/vb_$v.err; continue; }
  python - "$v" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/vb_{sys.argv[1]}.json"))
k = d["kernel_ms_per_frame"]; r = d["roofline"]
print(f"{sys.argv[1]:24s} frame {d['ms_per_step']:7.3f} ms  closest {k['ms_closest']:6.2f} (alone {r['alone']['avg_launch_ms'] * r['launches']:6.2f})  any {k['ms_any']:6.2f}  raygen {k['ms_raygen']:6.2f}  shade {k['ms_shade']:5.2f}")
PY
done
