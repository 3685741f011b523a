#!/bin/bash
# same-box A/B of environment switches on the search legs of the default bench, rotated order: tools/ab_search_env.sh R "VAR=a" "VAR=b" ...
R=$1; shift; cfgs=("$@"); n=${#cfgs[@]}
timeout -k 10 120 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-search > /dev/null 2>&1   # clocks
for ((r=0; r<R; r++)); do for ((i=0; i<n; i++)); do c=${cfgs[$(((i+r)%n))]}
  echo -n "$c: "
  env $c timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('search %.1f raxml-path %.1f gene-trees/s' % (d['search']['gene_trees_per_sec'], d['search_raxml_path']['gene_trees_per_sec']))"
done; done
