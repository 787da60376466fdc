#!/bin/bash
# A/B micro-benchmark of conv_pair kernel variants on the GPU box: tools/ab_pair.sh "case case ..." variant variant ...
CASES=$1; shift
for v in "$@"; do echo "== $v"; CPN_HIP_LIB=$PWD/celldetection_amd/build/variants/libcpn_$v.so python tools/pair_microbench.py $CASES 2>&1 | grep -v amdgpu.ids; done
