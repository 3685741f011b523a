#!/bin/bash
# same-box A/B of builds of libpeprml.so on the C3 chained scoring launch (k_oplist ms per launch + lnL of gene 0 as a check):
#   tools/ab_libs_score.sh NAME1 NAME2 ...      (build_ab/libpeprml_NAME.so)
cat > /tmp/abs.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from pepr_amd import engine, synth
genes = [synth.simulate_alignment(50, 1000, 1 + i, 0.8) for i in range(128)]
ctx = engine.Context(0, profile=True)
b = engine.Batch(ctx, [(g[0], g[1]) for g in genes], [g[2] for g in genes], alpha=0.8)
for _ in range(300): b.score()
ctx.kernel_stats(reset=True)
for _ in range(50): l = b.score()
st = ctx.kernel_stats()["newview"]
ctx.kernel_stats(reset=True)
for _ in range(50): l2 = b.score(stored=True)
st2 = ctx.kernel_stats()["newview"]
print("%-14s k_oplist %.4f ms/launch (stored %.4f)  lnL[0] %.9f %.9f" % (os.environ.get("TAG"), st["ms"] / st["launches"], st2["ms"] / st2["launches"], l[0], l2[0]))
PY
for rep in 1 2; do for a in "$@"; do
  set -- $a; env TAG="$a" PEPRML_LIB=$GRAFT_REPO_ROOT/build_ab/libpeprml_$1.so $2 timeout -k 10 120 python /tmp/abs.py 2>&1 | grep k_oplist
done; done
