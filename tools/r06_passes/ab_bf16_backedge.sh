#!/bin/bash
# Round 6: the bf16 main loops without the lgkmcnt(0) that closed every iteration (default) vs with it (be1: -DCPN_BACKEDGE_WAIT=1, rounds
# 2-5) vs with it pinned behind the last MFMA group (be2: -DCPN_BACKEDGE_WAIT=2)
cd "$GRAFT_REPO_ROOT"
D=$PWD/celldetection_amd/build/variants
CASES="k3 k5 k7 dec3 dec3b dec3cat head7 ref7 c64 bl7 pw1024 grp"
for rep in 1 2; do for z in 0 1; do
echo "== be0 zero=$z"; CPN_MB_ZERO=$z python tools/conv_microbench.py $CASES 2>&1 | grep -v amdgpu.ids
for v in be1 be2; do
echo "== $v zero=$z"; CPN_HIP_LIB=$D/libcpn_$v.so CPN_MB_ZERO=$z python tools/conv_microbench.py $CASES 2>&1 | grep -v amdgpu.ids
done; done; done
