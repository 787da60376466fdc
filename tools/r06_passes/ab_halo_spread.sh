#!/bin/bash
# Round 6: halo tile of the chunk after next requested piecewise over the steps of the running chunk (default) vs in one burst at the
# chunk change.  NEGATIVE (-1 ... -3 %): the piecewise issue (CPN_HALO_SPREAD) was removed from csrc/conv_igemm.hip again; this script and
# profiles/r06_ab_halo_spread.txt record the measurement.
cd "$GRAFT_REPO_ROOT"
D=$PWD/celldetection_amd/build/variants
CASES="k3 k5 k7 dec3 dec3b dec3cat head7 ref7 c64 bl7"
for rep in 1 2; do for z in 0 1; do for fp8 in 0 1; do
s=""; [ $fp8 = 1 ] && s=8
echo "== spread fp8=$fp8 zero=$z"; CPN_MB_ZERO=$z CPN_MB_FP8=$fp8 python tools/conv_microbench.py $CASES 2>&1 | grep -v amdgpu.ids
echo "== burst fp8=$fp8 zero=$z"; CPN_HIP_LIB=$D/libcpn_nospread$s.so CPN_MB_ZERO=$z CPN_MB_FP8=$fp8 python tools/conv_microbench.py $CASES 2>&1 | grep -v amdgpu.ids
done; done; done
