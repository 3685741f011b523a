#!/bin/bash
# same-box A/B of the chained kernel's variants (PML_CHAIN_VARIANT 11 = no prefetch, 10 = right side prefetched one category ahead)
timeout -k 10 120 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-search > /dev/null 2>&1   # clocks
arms=("PML_CHAIN_VARIANT=11" "PML_CHAIN_VARIANT=10")
for r in 0 1; do for i in 0 1; do a=${arms[$(((i+r)%2))]}
  echo -n "c3 $a: "
  env $a timeout -k 10 400 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.3f oplist %.4f search %.1f raxml-path %.1f gene-trees/s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['search']['gene_trees_per_sec'], d['search_raxml_path']['gene_trees_per_sec']))"
done; done
for r in 0 1; do for i in 0 1; do a=${arms[$(((i+r)%2))]}
  echo -n "c4 (24 genes) $a: "
  env $a timeout -k 10 600 python bench.py --workload c4 --genes 24 --scaling strong --steps 10 --warmup 3 --no-cpu-baseline --no-search 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.3f oplist %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
