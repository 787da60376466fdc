#!/bin/bash
# Round 6: cache policy of the halo (activation) LDS-DMA of the conv kernels + the traffic-free ablation (all halo lanes out of bounds:
# same instruction stream, no memory traffic).  Variants: tools/build_variant.sh NAME -DCPN_HALO_AUX=<2 nt | 3 sc0 nt | 18 sc1 nt | 19> /
# -DCPN_EXP_HALO_OOB=1 (and SRC=conv_fp8 ... NAME8)
cd "$GRAFT_REPO_ROOT"
D=$PWD/celldetection_amd/build/variants
CASES="k3 k5 k7 dec3 dec3b head7"
for fp8 in 0 1; do
s=""; [ $fp8 = 1 ] && s=8
echo "== default fp8=$fp8"; CPN_MB_FP8=$fp8 python tools/conv_microbench.py $CASES 2>&1 | grep -v amdgpu.ids
for v in oob nohdma nt a3 a18 a19; do
echo "== $v fp8=$fp8"; CPN_HIP_LIB=$D/libcpn_$v$s.so CPN_MB_FP8=$fp8 python tools/conv_microbench.py $CASES 2>&1 | grep -v amdgpu.ids
done
echo "== default again fp8=$fp8"; CPN_MB_FP8=$fp8 python tools/conv_microbench.py $CASES 2>&1 | grep -v amdgpu.ids
done
