#!/bin/bash
# per-bounce durations of the path shading kernel of two bench frames (rocprofv3 kernel trace, one frame at a time)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_sh -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --frames-in-flight 1 > gpurun_out/kt_sh.log 2>&1
python3 - <<'PY'
import csv,glob
t=glob.glob('gpurun_out/kt_sh/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(t)) if 'k_shade_path' in r['Kernel_Name']]
d=[round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows]
print("shade durations us:", d[8:24], "per frame ms", sum(d[8:16])/1e3)
PY
rm -rf gpurun_out/kt_sh
