cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "four_items or test_conv" 2>&1 | tail -6
for s in 0 1; do echo "### CPN_S1Q=$s"; CPN_S1Q=$s python tools/conv_microbench.py ref7 2>&1 | grep -v amdgpu.ids; done
timeout 600 python tools/ab_layers.py "CPN_S1Q=0" 2>&1 | grep -v amdgpu.ids
