cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
V=celldetection_amd/build/variants
for rep in 1 2; do
echo "## shipped (weights prefetched)"; python tools/bridge_microbench.py 2>&1 | grep bridge
echo "## -DCPN_BR_LATE_W"; CPN_HIP_LIB=$V/libcpn_br_latew.so python tools/bridge_microbench.py 2>&1 | grep bridge
echo "## -DCPN_BR_NOSTAGE1 (wrong results)"; CPN_HIP_LIB=$V/libcpn_br_nostage1.so python tools/bridge_microbench.py 2>&1 | grep bridge
echo "## -DCPN_EXP_NOEPI (wrong results)"; CPN_HIP_LIB=$V/libcpn_br_noepi.so python tools/bridge_microbench.py 2>&1 | grep bridge
done
