#!/bin/bash
# same-box A/B of the scoring kernel variants: k_oplist<11> (32 patterns per wave, 2 waves/SIMD) vs k_oplist16 (16, 4 waves/SIMD)
for rep in 1 2; do for v in 11 16; do
  PML_CHAIN_VARIANT=$v BENCH_NO_C4=1 BENCH_CLOCK_WARMUP_S=0.5 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-search > /tmp/ab.json 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python - <<PY
import json
d=json.loads([l for l in open("/tmp/ab.json") if l.startswith("{")][0])
print("variant $v: resident %.4f ms/launch (step %.4f, %.1f M site-lnL/s, frac %.3f), stored %.4f ms/launch" % (d["roofline"]["avg_launch_ms"], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["stored_traversal"]["roofline"]["avg_launch_ms"]))
PY
done; done
