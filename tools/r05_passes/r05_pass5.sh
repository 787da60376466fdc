cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_conv_bridge.py -m gpu -x -q 2>&1 | tail -25
timeout 600 python tools/ab_layers.py --fresh "CPN_BRIDGE=0" 2>&1 | grep -v amdgpu.ids
