#!/bin/bash
# usage (on the GPU box): tools/profile_cfg5.sh <tag>      -> gpurun_out/prof_<tag>_cfg5/ (copy what is to be judged into profiles/ as <tag>_cfg5_*)
# BASELINE config 5 (2048^2, 1 024 spp, depth 16, Plastic + Metal, four sphere area lights): the bench line, the kernel trace of one frame at a time,
# and the PMC passes behind the cfg5 line's `roofline` (TCP / TCC requests, FETCH_SIZE, WRITE_SIZE, SQ lane utilisation); one step = one frame of 16 passes
set -e
tag=$1
o=gpurun_out/prof_${tag}_cfg5
mkdir -p $o
python3 bench.py --config cfg5 --steps 4 --warmup 1 > $o/bench.log 2>&1
grep '^{' $o/bench.log | tail -1 > $o/bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace1 -- python3 bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 > $o/trace1.log 2>&1
grep '^{' $o/trace1.log | tail -1 > $o/bench_under_rocprof_1flight.json
cp $(find $o/trace1 -name "*kernel_stats.csv" | head -1) $o/kernel_stats_1flight.csv
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $o/pmc/pass$i -- python3 bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --frames-in-flight 1 > $o/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $o/pmc$i.log; }
done
python3 tools/pmc_traffic.py $o/pmc 512 > $o/pmc_traffic.json
python3 tools/pmc_kernels.py $o/pmc > $o/pmc_kernels.txt
rm -rf $o/trace1 $o/pmc   # (the raw per-dispatch files of 16 passes x 16 bounces per frame exceed what gpurun copies back: the summaries above are what is kept)
cat $o/bench.json; head -c 1500 $o/pmc_traffic.json; head -12 $o/pmc_kernels.txt | cut -c1-250
