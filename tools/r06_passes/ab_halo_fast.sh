#!/bin/bash
# Round 6: per-lane halo DMA offsets held in registers for MODE_S1 convs with one full-resolution source (default) vs the generic issue
# path.  NEGATIVE (bf16 +-0.5 %, e4m3 -2 ... -7 %, cycles per step up): the fast path (CPN_HALO_FAST, HFQ = 6 VGPRs of per-lane offsets filled from
# the column table, two scalar instructions per DMA at the chunk change) was removed from csrc/conv_igemm.hip again; this script and
# profiles/r06_ab_halo_fast.txt record the measurement.
cd "$GRAFT_REPO_ROOT"
D=$PWD/celldetection_amd/build/variants
CASES="k3 k5 k7 dec3 dec3b head7 ref7 c64"
for rep in 1 2; do for z in 0 1; do for fp8 in 0 1; do
s=""; [ $fp8 = 1 ] && s=8
echo "== fast fp8=$fp8 zero=$z"; CPN_MB_ZERO=$z CPN_MB_FP8=$fp8 python tools/conv_microbench.py $CASES 2>&1 | grep -v amdgpu.ids
echo "== generic fp8=$fp8 zero=$z"; CPN_HIP_LIB=$D/libcpn_nofast$s.so CPN_MB_ZERO=$z CPN_MB_FP8=$fp8 python tools/conv_microbench.py $CASES 2>&1 | grep -v amdgpu.ids
done; done; done
for z in 1 0; do for fp8 in 0 1; do s=""; [ $fp8 = 1 ] && s=8
for c in head7 k5 k3; do echo "== clockfast fp8=$fp8 zero=$z $c"; CPN_HIP_LIB=$D/libcpn_clockfast$s.so CPN_MB_ZERO=$z CPN_MB_FP8=$fp8 python tools/conv_microbench.py $c 2>&1 | grep CLK | tail -2; done; done; done
