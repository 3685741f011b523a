#!/bin/bash
# same-box A/B of register chaining (PML_CHAIN 0 off / 1 scoring passes / 2 every launch; PML_CHAIN_VARIANT 9 = two
# barriers per op, 11 = double-buffered LDS-DMA staging): C3 with the search legs, C4-shard scoring.  Rotated order.
timeout -k 10 120 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-search > /dev/null 2>&1   # clocks
arms=("PML_CHAIN=0" "PML_CHAIN=1 PML_CHAIN_VARIANT=9" "PML_CHAIN=1" "PML_CHAIN=2")
for r in 0 1; do for i in 0 1 2 3; do a=${arms[$(((i+r)%4))]}
  echo -n "c3 $a: "
  env $a timeout -k 10 400 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.3f oplist %.4f search %.1f raxml-path %.1f gene-trees/s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['search']['gene_trees_per_sec'], d['search_raxml_path']['gene_trees_per_sec']))"
done; done
arms=("PML_CHAIN=0" "PML_CHAIN=1 PML_CHAIN_VARIANT=9" "PML_CHAIN=1")
for r in 0 1; do for i in 0 1 2; do a=${arms[$(((i+r)%3))]}
  echo -n "c4 (24 genes) $a: "
  env $a timeout -k 10 600 python bench.py --workload c4 --genes 24 --scaling strong --steps 10 --warmup 3 --no-cpu-baseline --no-search 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.3f oplist %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
