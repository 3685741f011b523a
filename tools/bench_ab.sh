#!/bin/bash
# A/B of the traffic optimisations on the C3 bench (same box, same process order)
for cfg in "x=1" "PML_NO_PITCH=1" "PML_NO_CHERRY=1" "x=1" "PML_NO_PITCH=1"; do
  echo -n "$cfg: "
  env $cfg timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-search 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.2f M site-lnL/s  ms/step %.3f  oplist %.3f ms  pmat %.3f  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernels_ms_per_step']['pmat'], d['roofline']['frac']))"
done
