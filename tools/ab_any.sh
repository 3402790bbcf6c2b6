#!/bin/bash
# usage (GPU box): tools/ab_any.sh <variant|default>[:opt=value,...] ...   shadow launches on the main stream, one frame at a time: ms_any = the any-hit kernels alone
for spec in "$@"; do
  v=${spec%%:*}; opts="--opt overlap_shadow=0"; [ "$spec" != "$v" ] && for o in $(echo ${spec#*:} | tr ',' ' '); do opts="$opts --opt $o"; done
  lib=""; [ "$v" != "default" ] && lib=$PWD/build/variants/librrt_$v.so
  RRT_LIBRARY=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --frames-in-flight 1 $opts 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('%-36s frame %.3f  any alone %.3f  closest alone %.3f  shade %.3f  raygen %.3f' % ('$spec', d['ms_per_step'], k['ms_any'], k['ms_closest'], k['ms_shade'], k['ms_raygen']))"
done
