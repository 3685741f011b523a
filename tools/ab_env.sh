#!/bin/bash
# same-box A/B of environment switches on the scoring step, rotated order: tools/ab_env.sh R workload "VAR=a" "VAR=b" ...
R=$1; W=$2; shift 2; cfgs=("$@"); n=${#cfgs[@]}
timeout -k 10 120 python bench.py --workload $W --steps 200 --warmup 50 --no-cpu-baseline --no-search > /dev/null 2>&1   # clocks
for ((r=0; r<R; r++)); do for ((i=0; i<n; i++)); do c=${cfgs[$(((i+r)%n))]}
  echo -n "$c: "
  env $c timeout -k 10 400 python bench.py --workload $W --steps 40 --warmup 10 --no-cpu-baseline --no-search 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.3f  oplist %.4f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
