#!/bin/bash
# instruction-cache counters of the chained scoring kernel: tools/pmc_icache.sh
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/icache; mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $O/p1 --output-format csv -- python3 $R/tools/score_loop.py 12 > $O/p1.log 2>&1 || echo "pass failed"
cd $R && python tools/pmc_summary.py $O "k_oplist<11>" "icache" | head -40
