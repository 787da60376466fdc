#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "test_conv" --timeout=600 > gpurun_out/r2e_pytest_conv.log 2>&1; tail -3 gpurun_out/r2e_pytest_conv.log
for i in 1 2; do timeout 600 tools/ab.sh "head7 dec3b dec3 dec3cat c64 ref7 pw1024 pw256 grp" buffer kernarg; done > gpurun_out/r2e_ab.txt 2>&1; cat gpurun_out/r2e_ab.txt
timeout 600 python bench.py --no-cpu-baseline --profile-layers > gpurun_out/r2e_bench_n1.json 2> gpurun_out/r2e_per_layer_timing.txt; cut -c1-200 gpurun_out/r2e_bench_n1.json
