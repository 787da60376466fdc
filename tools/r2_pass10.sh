#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
CPN_HIP_LIB=$PWD/celldetection_amd/build/variants/libcpn_w44.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "test_conv" --timeout=500 2>&1 | tail -4
for i in 1 2; do timeout 600 tools/ab.sh "head7 dec3b dec3 dec3cat bl7 pw1024 pw256" base w44; done > gpurun_out/r2j_ab_w44.txt 2>&1; cat gpurun_out/r2j_ab_w44.txt
