#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
./tools/probes/buffer_lds_probe.bin > gpurun_out/r2c_buffer_probe.txt 2>&1; cat gpurun_out/r2c_buffer_probe.txt
timeout 900 python -m pytest tests/test_labels.py tests/test_preprocess.py tests/test_torch_ops.py -q -m gpu -s --timeout=600 > gpurun_out/r2c_pytest_new.log 2>&1; tail -25 gpurun_out/r2c_pytest_new.log
