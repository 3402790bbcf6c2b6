#!/bin/bash
# usage (on the GPU box): tools/fetch_calib.sh  -> gpurun_out/fetch_calib/summary.txt  (copy to profiles/r3_fetch_calibration.txt)
# FETCH_SIZE calibration on known-byte access patterns (tools/micro/fetch_calib.hip); PMC passes are separate and never combined with a trace.
set -e
o=gpurun_out/fetch_calib
mkdir -p $o build
hipcc -O3 --offload-arch=gfx950 tools/micro/fetch_calib.hip -o build/fetch_calib
./build/fetch_calib > $o/plain.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pass1 -- ./build/fetch_calib > $o/pass1.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $o/pass2 -- ./build/fetch_calib > $o/pass2.log 2>&1 || \
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $o/pass2 -- ./build/fetch_calib > $o/pass2.log 2>&1 || true
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $o/pass3 -- ./build/fetch_calib > $o/pass3.log 2>&1 || true
python3 tools/fetch_calib.py $o > $o/summary.txt
cat $o/summary.txt
