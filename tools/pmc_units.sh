#!/bin/bash
# usage (GPU box): [BENCH_OPTS="--opt k=v"] [RRT_LIBRARY=...] tools/pmc_units.sh <tag>   which unit of a CU the traversal kernels keep busy: LDS (bank conflicts, FIFO stalls),
# VALU / scalar / misc issue cycles, instruction fetch - two --pmc passes of a one-frame-at-a-time bench run -> gpurun_out/pu_<tag>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pu_$1; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_IFETCH SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --frames-in-flight 1 $BENCH_OPTS > $out.pass$i.log 2>&1 || { tail -5 $out.pass$i.log; exit 1; }
done
python3 - $out <<'PY' > gpurun_out/pu_$1.txt
import collections, csv, glob, re, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(f"{sys.argv[1]}/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void rrtd::", "").replace("rrtd::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0))[:7]:
    b = max(1.0, c.get("SQ_BUSY_CU_CYCLES", 0) / 2)    # two passes carry it
    print(k[:44])
    print("   per busy CU cycle: " + "  ".join(f"{n[3:]}={v / b:.3f}" for n, v in sorted(c.items()) if n != "SQ_BUSY_CU_CYCLES"))
PY
rm -rf $out
cat gpurun_out/pu_$1.txt | cut -c1-700
