#!/bin/bash
# Round 6: tile of the encoder's 1x1 convs (CPN_PW_TILE="TH,BN", read per call): the default heuristic's 8 x 32 px x 256 channels (one
# workgroup per CU, one round for the 1024-channel layers) against smaller tiles with more, co-resident workgroups
cd "$GRAFT_REPO_ROOT"
CASES="pw1024 pw512 pw256 pw2048"
for rep in 1 2; do for z in 0; do
echo "== default zero=$z"; CPN_MB_ZERO=$z python tools/conv_microbench.py $CASES 2>&1 | grep -v amdgpu.ids
for t in 8,128 4,256 4,128 4,64 8,64; do
echo "== tile $t zero=$z"; CPN_PW_TILE=$t CPN_MB_ZERO=$z python tools/conv_microbench.py $CASES 2>&1 | grep -v amdgpu.ids
done; done; done
