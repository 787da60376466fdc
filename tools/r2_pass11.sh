#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
CPN_HIP_LIB=$PWD/celldetection_amd/build/variants/libcpn_xcd.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "test_conv" --timeout=500 2>&1 | tail -2
for i in 1 2; do tools/ab.sh "head7 dec3b dec3 dec3cat c64 ref7 pw1024 pw256" base xcd; done
echo "== tile env (pw1024 / pw256 with 4,128 | 4,256 | 8,128)"
for t in 8,256 4,256 8,128 4,128; do echo "-- CPN_TILE=$t"; CPN_TILE=$t CPN_HIP_LIB=$PWD/celldetection_amd/build/variants/libcpn_tile.so python tools/conv_microbench.py pw1024 pw256 dec3 2>&1 | grep -v amdgpu; done
