#!/bin/bash
# same-box A/B of the scoring step for several builds, rotated order, R rounds: tools/ab_score.sh R workload lib1 lib2 ...
R=$1; W=$2; shift 2; libs=("$@"); n=${#libs[@]}
timeout -k 10 120 python bench.py --workload $W --steps 200 --warmup 50 --no-cpu-baseline --no-search > /dev/null 2>&1   # clocks
for ((r=0; r<R; r++)); do for ((i=0; i<n; i++)); do lib=${libs[$(((i+r)%n))]}
  echo -n "$lib: "
  PEPRML_LIB=$PWD/$lib timeout -k 10 400 python bench.py --workload $W --steps 40 --warmup 10 --no-cpu-baseline --no-search 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.3f  oplist %.4f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
