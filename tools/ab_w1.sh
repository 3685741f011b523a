#!/bin/bash
# same-box probe: the chained scoring kernel compiled for ONE wave per SIMD (512 registers) against the product build
# (two waves per SIMD): tools/ab_w1.sh  (arms built in the container)
cat > /tmp/w1.py <<'PY'
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
print("%-22s k_oplist %.4f ms/launch  lnL[0] %.6f" % (os.environ.get("TAG"), st["ms"] / st["launches"], l[0]))
PY
for rep in 1 2; do for a in BASE W1 W1F2 "W1 PML_CHAIN_VARIANT=10" "BASE PML_CHAIN_VARIANT=10"; do
  set -- $a; lib=$GRAFT_REPO_ROOT/build_ab/libpeprml_$1.so
  env TAG="$a" PEPRML_LIB=$lib $2 timeout -k 10 120 python /tmp/w1.py 2>&1 | grep k_oplist
done; done
