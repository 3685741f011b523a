#!/bin/bash
# same-box A/B of builds of libpeprml.so on the C3 scoring pass + searches: tools/ab_two_libs.sh libA.so libB.so ...
for rep in 1 2; do for lib in "$@"; do
  PEPRML_LIB=$PWD/$lib BENCH_NO_C4=1 BENCH_CLOCK_WARMUP_S=0.5 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /tmp/ab.json 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python - <<PY
import json
d=json.loads([l for l in open("/tmp/ab.json") if l.startswith("{")][0])
print("$lib: resident %.4f ms/launch (%.1f M site-lnL/s), stored %.4f ms/launch; search NNI %.1f RAxML %.1f gene-trees/s" % (d["roofline"]["avg_launch_ms"], d["value"], d["stored_traversal"]["roofline"]["avg_launch_ms"], d["search"]["gene_trees_per_sec"], d["search_raxml_path"]["gene_trees_per_sec"]))
PY
done; done
