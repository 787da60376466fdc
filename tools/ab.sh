#!/bin/bash
# A/B micro-benchmark of kernel variants on the GPU box: tools/ab.sh "case case ..." variant variant ...
CASES=$1; shift
for v in "$@"; do echo "== $v"; CPN_HIP_LIB=$PWD/celldetection_amd/build/variants/libcpn_$v.so python tools/conv_microbench.py $CASES 2>&1 | grep -v amdgpu.ids; done
