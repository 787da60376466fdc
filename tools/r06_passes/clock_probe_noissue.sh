cd "$GRAFT_REPO_ROOT"
D=$PWD/celldetection_amd/build/variants
for z in 1; do for fp8 in 0 1; do
s=""; [ $fp8 = 1 ] && s=8
for v in clock clocknoi clocknoh; do for c in head7 k5 k3; do
echo "== $v fp8=$fp8 zero=$z $c"; CPN_HIP_LIB=$D/libcpn_$v$s.so CPN_MB_ZERO=$z CPN_MB_FP8=$fp8 python tools/conv_microbench.py $c 2>&1 | grep -v amdgpu.ids | tail -3
done; done; done; done
