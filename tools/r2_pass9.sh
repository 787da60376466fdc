#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -q --timeout=900 > gpurun_out/r2i_pytest.log 2>&1; tail -3 gpurun_out/r2i_pytest.log
timeout 600 python bench.py --model CpnResNet50FPN --batch 8 --tile 1024 --precision fp8 --no-cpu-baseline --steps 5 --warmup 2 --profile-layers > gpurun_out/r2i_bench_cfg4_fp8.json 2> gpurun_out/r2i_cfg4_fp8_layers.txt; cut -c1-200 gpurun_out/r2i_bench_cfg4_fp8.json; tail -6 gpurun_out/r2i_cfg4_fp8_layers.txt
