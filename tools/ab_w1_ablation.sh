#!/bin/bash
# timing-only ablations (results are NOT likelihoods) of the chained scoring kernel at ONE wave per SIMD, where stalls do not overlap
# with another wave's work and so add up: tools/ab_w1_ablation.sh   (arms: build_ab/libpeprml_<ARM>.so, built in the container)
cat > /tmp/w1a.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from pepr_amd import engine, synth
genes = [synth.simulate_alignment(50, 1000, 1 + i, 0.8) for i in range(128)]
ctx = engine.Context(0, profile=True)
b = engine.Batch(ctx, [(g[0], g[1]) for g in genes], [g[2] for g in genes], alpha=0.8)
def sc():
    try: b.score()
    except Exception: pass
for _ in range(300): sc()
ctx.kernel_stats(reset=True)
for _ in range(50): sc()
st = ctx.kernel_stats()["newview"]
print("%-14s k_oplist %.4f ms/launch over %d launches" % (os.environ.get("TAG"), st["ms"] / max(st["launches"], 1), st["launches"]))
PY
for rep in 1 2; do for a in BASE W2_NO_OP W1_NO_OP; do
  TAG=$a PEPRML_LIB=$GRAFT_REPO_ROOT/build_ab/libpeprml_$a.so timeout -k 10 120 python /tmp/w1a.py 2>&1 | grep k_oplist
done; done
