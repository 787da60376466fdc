cd "$GRAFT_REPO_ROOT"
D=$PWD/celldetection_amd/build/variants
for rep in 1 2; do
echo "== fp8 default"; CPN_MB_FP8=1 python tools/conv_microbench.py dec3b dec3 k3 k5 2>&1 | grep -v amdgpu.ids
echo "== fp8 S1F enabled"; CPN_HIP_LIB=$D/libcpn_fp8s1f.so CPN_MB_FP8=1 python tools/conv_microbench.py dec3b dec3 k3 k5 2>&1 | grep -v amdgpu.ids
done
CPN_HIP_LIB=$D/libcpn_fp8s1f.so timeout 900 python -m pytest tests -q -m gpu -k "fp8" 2>&1 | tail -3
timeout 600 python bench.py --precision fp8 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-160
CPN_HIP_LIB=$D/libcpn_fp8s1f.so timeout 600 python bench.py --precision fp8 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-160
