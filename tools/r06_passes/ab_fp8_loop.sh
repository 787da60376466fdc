#!/bin/bash
# Round 6: A/B of the e4m3 main loop of csrc/conv_igemm.hip on one MI355X box (tools/conv_microbench.py, random and all-zero operands).
#   default build            whole operand sets + alternating wave priority
#   noprio                   whole operand sets only            (SRC=conv_fp8 tools/build_variant.sh noprio -DCPN_WAVE_PRIO=0)
#   halfsets                 the loop of rounds 3-5            (SRC=conv_fp8 tools/build_variant.sh halfsets -DCPN_FP8_HALFSETS -DCPN_WAVE_PRIO=0)
cd "$GRAFT_REPO_ROOT"
D=$PWD/celldetection_amd/build/variants
for rep in 1 2; do for z in 0 1; do
echo "== default zero=$z"; CPN_MB_FP8=1 CPN_MB_ZERO=$z python tools/conv_microbench.py head7 k5 dec3b k3 2>&1 | grep -v amdgpu.ids
for v in noprio halfsets; do
echo "== $v zero=$z"; CPN_HIP_LIB=$D/libcpn_$v.so CPN_MB_FP8=1 CPN_MB_ZERO=$z python tools/conv_microbench.py head7 k5 dec3b k3 2>&1 | grep -v amdgpu.ids
done; done; done
