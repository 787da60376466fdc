#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python tools/conv_microbench.py bl7 bl7s head7 2>&1 | grep -v amdgpu.ids
CPN_MB_FP8=1 python tools/conv_microbench.py bl7 bl7s head7 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "test_conv" --timeout=600 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py -q -k "config4 or config1 or fp8_precision or conv_stack" --timeout=800 2>&1 | tail -2
