#!/bin/bash
# Round 6, VERDICT r5 item 7: the encoder's 1x1 family with operands loaded straight into registers.  What exists is MODE_PWR
# (CPN_PWR=1: the WEIGHT operand of 1x1 convs from L2 into registers, no weight tiles in LDS, half the LDS-DMA instructions);
# this pass records it against the default LDS-DMA loop on the 1x1 shapes of the ResNeXt101 encoder / decoder, same box.
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for p in 0 1; do
echo "== CPN_PWR=$p"; CPN_PWR=$p python tools/conv_microbench.py pw256 pw512 pw1024 pw2048 2>&1 | grep -v amdgpu.ids
done; done
