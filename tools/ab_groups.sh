#!/bin/bash
# same-box A/B of the number of groups a search call is dealt over (PML_GROUPS): tools/ab_groups.sh OUT
O=gpurun_out/$1; mkdir -p $O
for g in 2 4 3 4 2 3; do
  PML_GROUPS=$g BENCH_NO_C4=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/g$g.json 2> $O/g$g.err || echo "groups $g failed"
  python - <<PY
import json
d=json.loads([l for l in open("$O/g$g.json") if l.startswith("{")][0])
print("groups $g: NNI %.1f gene-trees/s (%.3f s, cold %.3f s)  RAxML path %.1f (%.3f s)  fallbacks %s" % (d["search"]["gene_trees_per_sec"], d["search"]["seconds"], d["search"]["cold_first_call_seconds"], d["search_raxml_path"]["gene_trees_per_sec"], d["search_raxml_path"]["seconds"], d["search"]["newton_fallbacks_rank0"]))
PY
done
