#!/bin/bash
# per-bounce kernel durations of one bench frame (rocprofv3 kernel trace)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fp_$tag -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/fp_$tag.log 2>&1
grep '^{' gpurun_out/fp_$tag.log | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('value',d['value'],'ms',d['ms_per_step'],'roof',d['roofline']['achieved'],d['roofline']['frac'],d['kernel_ms_per_frame'])"
t=$(find gpurun_out/fp_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
def dur(r): return (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
import re
groups=collections.OrderedDict()
for r in rows:
    n=r["Kernel_Name"]
    key=re.sub(r"\(.*","",n).replace("void rrtd::","")
    groups.setdefault(key,[]).append(dur(r))
for k,v in groups.items():
    if len(v)>=8 and ("trace" in k or "k_trace" in k or "closest" in k or "shadow" in k or "shade_path" in k):
        per=[0]*8; cnt=[0]*8
        for i,x in enumerate(v): per[i%8]+=x; cnt[i%8]+=1
        print(k[:40], "calls",len(v),"total ms %.1f"%(sum(v)/1e3), "per-bounce mean us", [round(per[i]/max(1,cnt[i])) for i in range(8)])
    else:
        print(k[:40], "calls",len(v),"total ms %.1f"%(sum(v)/1e3), "mean us %.1f"%(sum(v)/len(v)))
PY
