#!/bin/bash
# timing-only ablations of k_oplist<11> (results are NOT valid likelihoods): which source of stalls is worth how much.
# Build the arms first, in the container (hipcc cross-compiles; the .so files travel with gpurun):
#   for a in NO_ROWS NO_CLV NO_LDS NO_BARRIER; do (cd pepr_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DABL_$a -x hip \
#       kernels.hip parsimony.hip engine.cpp host.cpp api.cpp search.cpp jackknife.cpp -shared -pthread -o ../../build_ab/libpeprml_$a.so); done
cat > /tmp/abl.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from pepr_amd import engine, synth
genes = [synth.simulate_alignment(50, 1000, 1 + i, 0.8) for i in range(128)]
ctx = engine.Context(0, profile=True)
b = engine.Batch(ctx, [(g[0], g[1]) for g in genes], [g[2] for g in genes], alpha=0.8)
try:
    for _ in range(300): b.score()
except Exception as e: pass
ctx.kernel_stats(reset=True)
n = 0
for _ in range(50):
    try: b.score(); n += 1
    except Exception: pass
st = ctx.kernel_stats()["newview"]
print("%-12s k_oplist %.4f ms/launch over %d launches" % (os.environ.get("TAG"), st["ms"] / max(st["launches"], 1), st["launches"]))
PY
for rep in 1 2; do for a in BASE NO_ROWS NO_CLV NO_LDS NO_BARRIER; do
  if [ $a = BASE ]; then lib=$GRAFT_REPO_ROOT/pepr_amd/libpeprml.so; else lib=$GRAFT_REPO_ROOT/build_ab/libpeprml_$a.so; fi
  TAG=$a PEPRML_LIB=$lib timeout -k 10 120 python /tmp/abl.py 2>&1 | grep k_oplist
done; done
