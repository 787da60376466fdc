#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
TAG=${1:-r2b}
timeout 1800 python -m pytest tests -q -m gpu --timeout=900 -s > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -15 gpurun_out/${TAG}_pytest_gpu.log
