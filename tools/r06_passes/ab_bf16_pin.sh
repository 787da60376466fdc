cd "$GRAFT_REPO_ROOT"
D=$PWD/celldetection_amd/build/variants
for rep in 1 2; do for z in 0 1; do
echo "== bf16 default zero=$z"; CPN_MB_ZERO=$z python tools/conv_microbench.py head7 k5 dec3b dec3 dec3cat 2>&1 | grep -v amdgpu.ids
echo "== bf16 pinned zero=$z"; CPN_HIP_LIB=$D/libcpn_pinbf16.so CPN_MB_ZERO=$z python tools/conv_microbench.py head7 k5 dec3b dec3 dec3cat 2>&1 | grep -v amdgpu.ids
done; done
CPN_HIP_LIB=$D/libcpn_pinbf16.so timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "conv" 2>&1 | tail -2
