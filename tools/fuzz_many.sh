#!/bin/bash
# usage (GPU box): tools/fuzz_many.sh <first_seed> <n_seeds> <cases_per_seed>   -> gpurun_out/fuzz_<seed>.log, one summary line per seed
# odd seeds run the f64 mode, even seeds the fp32 mode; every third seed the deep DirectLighting / Debug bias
mkdir -p gpurun_out
for ((s = $1; s < $1 + $2; s++)); do
  mode=""; [ $((s % 2)) -eq 0 ] && mode="f32"
  deep=0; [ $((s % 3)) -eq 0 ] && deep=1
  FUZZ_DEEP_DIRECT=$deep timeout -k 10 600 python tools/fuzz_parity.py $3 $s $mode > gpurun_out/fuzz_$s.log 2>&1 || { echo "seed $s: fuzz_parity failed (rc $?)"; tail -5 gpurun_out/fuzz_$s.log; exit 1; }
  echo "seed $s ${mode:-f64} deep=$deep: ok $(grep -c '^ok' gpurun_out/fuzz_$s.log) diff $(grep -c '^DIFF' gpurun_out/fuzz_$s.log) non-cfg1 diff $(grep '^DIFF' gpurun_out/fuzz_$s.log | grep -vc ' cfg1 ') mismatch $(grep -c '^MISMATCH' gpurun_out/fuzz_$s.log)"
done
