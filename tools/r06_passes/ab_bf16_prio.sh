cd "$GRAFT_REPO_ROOT"
D=$PWD/celldetection_amd/build/variants
for rep in 1 2; do for z in 0 1; do
echo "== bf16 default zero=$z"; CPN_MB_ZERO=$z python tools/conv_microbench.py head7 k5 dec3b dec3 2>&1 | grep -v amdgpu.ids
for v in bprio1 bprio5; do
echo "== $v zero=$z"; CPN_HIP_LIB=$D/libcpn_$v.so CPN_MB_ZERO=$z python tools/conv_microbench.py head7 k5 dec3b dec3 2>&1 | grep -v amdgpu.ids
done; done; done
