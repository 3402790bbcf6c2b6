#!/bin/bash
# usage (GPU box): [BENCH_OPTS="--opt key=value ..."] tools/pmc_variant.sh <name>[:opt=value,...] [...]   ("default" = rs_ray_toy_amd/csrc/librrt.so)
# SQ and TCP / TCC counters (three --pmc passes) of a short one-frame-at-a-time bench.py run per kernel-tuning variant -> gpurun_out/pv_<name>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "$@"; do   # <variant|default>[:opt=value,...]
  v=${spec%%:*}; opts=""; [ "$spec" != "$v" ] && for o in $(echo ${spec#*:} | tr ',' ' '); do opts="$opts --opt $o"; done
  lib=""; [ "$v" != "default" ] && lib=$PWD/build/variants/librrt_$v.so
  export RRT_LIBRARY=$lib
  v=$(echo $spec | tr ':=,' '___')
  out=gpurun_out/pv_$v; rm -rf $out; mkdir -p $out
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --frames-in-flight 1 $BENCH_OPTS $opts > $out.pass$i.log 2>&1 || { tail -5 $out.pass$i.log; exit 1; }
  done
  python3 tools/pmc_kernels.py $out > gpurun_out/pv_$v.txt
  python3 - $out <<'PY' >> gpurun_out/pv_$v.txt
import collections, csv, glob, re, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{sys.argv[1]}/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void rrtd::", "").replace("rrtd::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:8]:
    print(k[:40], " ".join(f"{n[3:]}={v:.4g}" for n, v in sorted(c.items())))
PY
  rm -rf $out/pass*
  echo "== $v"; head -6 gpurun_out/pv_$v.txt | cut -c1-200
done
