#!/bin/bash
# same-box A/B of k_pmat builds on the C3 scoring step: tools/ab_pmat.sh R lib1 lib2 ...
R=$1; shift; libs=("$@"); n=${#libs[@]}
timeout -k 10 120 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-search > /dev/null 2>&1   # clocks
for ((r=0; r<R; r++)); do for ((i=0; i<n; i++)); do lib=${libs[$(((i+r)%n))]}
  echo -n "$lib: "
  PEPRML_LIB=$PWD/$lib timeout -k 10 400 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-search 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.3f  pmat %.4f  oplist %.4f ms' % (d['ms_per_step'], d['kernels_ms_per_step']['pmat'], d['roofline']['avg_launch_ms']))"
done; done
