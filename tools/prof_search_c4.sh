#!/bin/bash
# rocprofv3 kernel trace of one C4-shaped NNI search: tools/prof_search_c4.sh OUT NGENES
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
NO_PROFILE=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gpu_search_time.py $2 200 5000 0 > $O/search.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv 2>/dev/null
grep -v amdgpu.ids $O/search.log | grep -v "^[WE]20" | tail -12; head -8 $O/kernel_stats.csv | cut -c1-170
