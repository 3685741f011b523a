#!/bin/bash
# rocprofv3 kernel trace of one C3 NNI search (128 genes): tools/prof_search.sh OUT [env assignments...]
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
NO_PROFILE=1 timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_search_time.py 128 50 1000 0 > $O/search.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv 2>/dev/null
grep -v amdgpu.ids $O/search.log | tail -8; head -8 $O/kernel_stats.csv | cut -c1-160
