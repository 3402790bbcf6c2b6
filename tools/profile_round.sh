#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/ (copy what is to be judged into profiles/)
# 1. default bench.py (JSON line)
# 2. rocprofv3 --kernel-trace --stats of the same command (no cpu baseline): kernel_stats.csv
# 3. the same with one frame at a time (--frames-in-flight 1): kernel_stats_1flight.csv (no launch waits behind the other frame's kernels)
# 4. five separate --pmc passes (never combined with a trace): SQ lane utilisation / waits, SQ instruction mix, TCP / TCC hits, FETCH_SIZE,
#    WRITE_SIZE -> pmc_traffic.json (bytes per frame per kernel family + VALU lane utilisation, tied to the kernel sources by a hash)
#    and pmc_kernels.txt (per-kernel summary of all counters)
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
rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace1 -- python3 bench.py --no-cpu-baseline --frames-in-flight 1 > $o/trace1.log 2>&1
grep '^{' $o/trace1.log | tail -1 > $o/bench_under_rocprof_1flight.json
cp $(find $o/trace1 -name "*kernel_stats.csv" | head -1) $o/kernel_stats_1flight.csv
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $o/pmc/pass$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --frames-in-flight 1 > $o/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $o/pmc$i.log; }
done
python3 tools/pmc_traffic.py $o/pmc > $o/pmc_traffic.json
python3 tools/pmc_kernels.py $o/pmc > $o/pmc_kernels.txt
rm -rf $o/trace/*/*kernel_trace.csv $o/trace1/*/*kernel_trace.csv $o/pmc/pass*/*/*agent_info.csv
cat $o/bench.json; cat $o/pmc_traffic.json; cat $o/pmc_kernels.txt
