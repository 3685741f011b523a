#!/bin/bash
# round-end measurements in one box visit (tests are run separately): default bench line, rocprofv3 kernel stats of the scoring
# bench, PMC passes for the chained and the stored traversal (c3) and the traffic passes (c3, c4):  tools/final_measure.sh TAG
T=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 500 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench rc=$?"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${T}_stats --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-search --no-c4 > $O/${T}_bench_under_rocprof.json 2> $O/${T}_rocprof.err )
cp $(ls $O/${T}_stats/*/*kernel_stats.csv | head -1) $O/${T}_kernel_stats.csv; rm -rf $O/${T}_stats
bash tools/pmc_collect.sh gpurun_out/${T}_pmc_chained c3 --no-stored; python tools/pmc_summary.py $O/${T}_pmc_chained "k_oplist<11>" "c3 (128 genes x 50 taxa x 1000 sites): the register-resident scoring pass (chained children neither written nor read back), per launch" > $O/${T}_pmc_chained.json
bash tools/pmc_collect.sh gpurun_out/${T}_pmc_stored c3 --stored-only; python tools/pmc_summary.py $O/${T}_pmc_stored "k_oplist<11>" "c3: the STORED traversal (every CLV written, children read back unless chained), per launch" > $O/${T}_pmc_stored.json
PMC_PASSES=2 bash tools/pmc_collect.sh gpurun_out/${T}_pmc_c4 c4 --no-stored
find $O -name "*agent_info.csv" -delete
ls $O | grep ${T}; head -4 $O/${T}_kernel_stats.csv | cut -c1-150; tail -1 $O/${T}_bench.json | cut -c1-600
