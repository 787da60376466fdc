cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_conv_bridge.py tests/test_preprocess.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2; do
echo "## MODE_BRF (two workgroups per CU)"; python tools/bridge_microbench.py 2>&1 | grep bridge
echo "## CPN_BRF=0 (MODE_BR, one workgroup per CU)"; CPN_BRF=0 python tools/bridge_microbench.py 2>&1 | grep bridge
done
timeout 600 python tools/ab_layers.py "CPN_BRF=0" 2>&1 | grep -v amdgpu.ids
