#!/bin/bash
# how often does the full parity file end with a coalesced search that is not bit-identical to its lone call?
# tools/repro_rate.sh RUNS OUTFILE     (one pytest process per run, sequential)
R=$1; OUT=$2; : > $OUT
for ((r=0; r<R; r++)); do
  timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -q -s -p no:cacheprovider 2>&1 | grep -E "not bit-identical|passed|failed" | tr '\n' ' ' >> $OUT
  echo >> $OUT
done
echo "runs with a difference: $(grep -c 'identical to the lone calls: [1-8]' $OUT) of $R" | tee -a $OUT
