#!/bin/bash
# runs the C3 bench with each k_oplist variant (PML_OPLIST_VARIANT) -- kernel tuning aid
for c in 0 1; do for v in 1 5 0 3; do
  if [ $c = 1 ]; then export PML_NO_CHERRY=1; else unset PML_NO_CHERRY; fi
  echo -n "no_cherry=$c variant $v: "
  PML_OPLIST_VARIANT=$v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-search 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.2f M site-lnL/s  ms/step %.3f  oplist %.3f ms  %.0f GB/s frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['achieved'], d['roofline']['frac']))"
done; done
