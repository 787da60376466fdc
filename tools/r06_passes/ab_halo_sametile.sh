cd "$GRAFT_REPO_ROOT"
D=$PWD/celldetection_amd/build/variants
CASES="k3 k5 k7 dec3 dec3b head7"
for z in 1 0; do for fp8 in 0 1; do
s=""; [ $fp8 = 1 ] && s=8
for v in nospread sametile nohdma; do
echo "== $v fp8=$fp8 zero=$z"; CPN_HIP_LIB=$D/libcpn_$v$s.so CPN_MB_ZERO=$z CPN_MB_FP8=$fp8 python tools/conv_microbench.py $CASES 2>&1 | grep -v amdgpu.ids
done; done; done
