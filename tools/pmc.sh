#!/bin/bash
# usage: tools/pmc.sh <outdir> <program args...>   (runs separate --pmc passes; summaries: tools/pmc_kernels.py <outdir> and tools/pmc_traffic.py <outdir>)
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- "$@" > $out.pass$i.log 2>&1 || { tail -5 $out.pass$i.log; exit 1; }
done
