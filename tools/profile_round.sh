#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag>
# 1. default bench.py (JSON line)  2. rocprofv3 --kernel-trace --stats of the same command (no cpu baseline)
# 3. separate --pmc passes for FETCH_SIZE / WRITE_SIZE (HBM traffic of the traversal kernels)
set -e
tag=$1
o=gpurun_out/prof_$tag
mkdir -p $o
python3 bench.py > $o/bench.log 2>&1
grep '^{' $o/bench.log | tail -1 > $o/bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- python3 bench.py --no-cpu-baseline > $o/trace.log 2>&1
grep '^{' $o/trace.log | tail -1 > $o/bench_under_rocprof.json
cp $(find $o/trace -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc/pass1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --frames-in-flight 1 > $o/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc/pass2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --frames-in-flight 1 > $o/pmc2.log 2>&1
python3 tools/pmc_traffic.py $o/pmc > $o/pmc_traffic.json
rm -rf $o/trace/*/*kernel_trace.csv
cat $o/bench.json; cat $o/pmc_traffic.json
