#!/bin/bash
# per-operation cycle counts of the chained scoring kernel, one and two waves per SIMD: tools/optime.sh   (arms built in the container:
#   hipcc ... -DPML_OPTIME [-DPML_CHAIN_WAVES=1] ... -o build_ab/libpeprml_W{2,1}_OPTIME.so)
for a in W2_OPTIME W1_OPTIME; do TAG=$a PEPRML_LIB=$GRAFT_REPO_ROOT/build_ab/libpeprml_$a.so timeout -k 10 200 python tools/optime.py 2>&1 | grep -v amdgpu.ids; done
