# round 6 A/B of the e4m3 main loop (tools/conv_microbench.py, random and all-zero operands):
#   default build = whole operand sets + staggered boundary DMA; nostagger = whole sets only; halfsets = the round 3-5 loop
cd $GRAFT_REPO_ROOT
D=$PWD/celldetection_amd/build/variants
for rep in 1 2; do
for z in 0 1; do
echo "== stagger (default) zero=$z"; CPN_MB_FP8=1 CPN_MB_ZERO=$z python tools/conv_microbench.py head7 k5 dec3b k3 2>&1 | grep -v amdgpu.ids
echo "== nostagger zero=$z"; CPN_HIP_LIB=$D/libcpn_nostagger.so CPN_MB_FP8=1 CPN_MB_ZERO=$z python tools/conv_microbench.py head7 k5 dec3b k3 2>&1 | grep -v amdgpu.ids
echo "== halfsets (round 5) zero=$z"; CPN_HIP_LIB=$D/libcpn_halfsets.so CPN_MB_FP8=1 CPN_MB_ZERO=$z python tools/conv_microbench.py head7 k5 dec3b k3 2>&1 | grep -v amdgpu.ids
done; done
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "fp8" 2>&1 | tail -3
