#!/bin/bash
# same-box A/B of the virtual-node switches on the C3 scoring pass: tools/ab_virtual.sh
for rep in 1 2; do for e in "X=1" "PML_NO_PITCH=1" "PML_NO_CHERRY=1"; do
  env $e BENCH_NO_C4=1 BENCH_CLOCK_WARMUP_S=0.5 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-search > /tmp/ab.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open("/tmp/ab.json") if l.startswith("{")][0])
print("$e: resident %.4f ms/launch (step %.4f), stored %.4f ms/launch; pmat %.4f" % (d["roofline"]["avg_launch_ms"], d["ms_per_step"], d["stored_traversal"]["roofline"]["avg_launch_ms"], d["kernels_ms_per_step"]["pmat"]))
PY
done; done
