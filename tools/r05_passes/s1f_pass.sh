cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv or subpixel" 2>&1 | tail -8
for s in 0 1; do echo "### CPN_S1F=$s"; CPN_S1F=$s python tools/conv_microbench.py k3 k5 dec3 dec3b dec3cat 2>&1 | grep -v amdgpu.ids; done
python tools/ab_layers.py "CPN_S1F=0" 2>&1 | grep -v amdgpu.ids
