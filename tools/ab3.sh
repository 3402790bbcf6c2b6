#!/bin/bash
# repeated A/B: tools/ab3.sh <n> spec...
n=$1; shift
for i in $(seq 1 $n); do for spec in "$@"; do
  v=${spec%%:*}; opts=""; [ "$spec" != "$v" ] && for o in $(echo ${spec#*:} | tr ',' ' '); do opts="$opts --opt $o"; done
  lib=""; [ "$v" != "default" ] && lib=$PWD/build/variants/librrt_$v.so
  RRT_LIBRARY=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 12 --warmup 2 $opts 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$spec', d['ms_per_step'], d['roofline']['alone']['avg_launch_ms']*8)"
done; done
