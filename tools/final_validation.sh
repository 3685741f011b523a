#!/bin/bash
# round-end evidence in one box visit: GPU tests, smoke, the default bench line, rocprofv3 kernel stats of the scoring bench,
# the two PMC traffic passes.  tools/final_validation.sh TAG   (writes gpurun_out/TAG_*)
T=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 800 python -u -m pytest tests -m gpu -q > $O/${T}_tests.log 2>&1; tail -2 $O/${T}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1; tail -1 $O/${T}_smoke.log
timeout -k 10 400 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${T}_stats --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-search > $O/${T}_bench_under_rocprof.json 2> $O/${T}_rocprof.err
cp $(ls $O/${T}_stats/*/*kernel_stats.csv | head -1) $O/${T}_kernel_stats.csv; rm -rf $O/${T}_stats
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c -d $O/${T}_pmc/$c --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-search > $O/${T}_pmc_$c.log 2>&1
done
cd $R; python tools/pmc_summary.py $O/${T}_pmc > $O/${T}_pmc_summary.json; rm -rf $O/${T}_pmc
head -4 $O/${T}_kernel_stats.csv; cat $O/${T}_pmc_summary.json
