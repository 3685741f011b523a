#!/bin/bash
# same-box A/B of two builds of libpeprml.so: scoring step (bench) and the C3 / C4-shard searches
# usage: tools/ab_lib.sh pepr_amd/libpeprml_old.so pepr_amd/libpeprml.so
for round in 1 2; do for lib in "$@"; do
  echo "== $lib (round $round)"
  PEPRML_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-search 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  score: %.2f M site-lnL/s  ms/step %.3f  oplist %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
  PEPRML_LIB=$PWD/$lib NO_PROFILE=1 timeout -k 10 300 python tools/gpu_search_time.py 128 50 1000 0 2>&1 | grep "search 128" | sed 's/^/  /'
done; done
for lib in "$@"; do
  echo "== $lib C4 shard"
  PEPRML_LIB=$PWD/$lib timeout -k 10 400 python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline --no-search 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  score: %.2f M site-lnL/s  ms/step %.3f  oplist %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
