#!/bin/bash
# Round 6: what the per-chunk halo staging costs by tap count (structural ablation -DCPN_EXP_NOHDMA: steady-state halo DMA skipped,
# wrong results by construction).   tools/build_variant.sh nohdma -DCPN_EXP_NOHDMA ; SRC=conv_fp8 tools/build_variant.sh nohdma8 -DCPN_EXP_NOHDMA
cd "$GRAFT_REPO_ROOT"
D=$PWD/celldetection_amd/build/variants
for z in 1 0; do
for fp8 in 0 1; do
v=nohdma; [ $fp8 = 1 ] && v=nohdma8
echo "== default fp8=$fp8 zero=$z"; CPN_MB_FP8=$fp8 CPN_MB_ZERO=$z python tools/conv_microbench.py k3 k5 k7 dec3 dec3b 2>&1 | grep -v amdgpu.ids
echo "== nohdma fp8=$fp8 zero=$z"; CPN_HIP_LIB=$D/libcpn_$v.so CPN_MB_FP8=$fp8 CPN_MB_ZERO=$z python tools/conv_microbench.py k3 k5 k7 dec3 dec3b 2>&1 | grep -v amdgpu.ids
done; done
