#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "test_conv" --timeout=600 > gpurun_out/r2f_pytest_conv.log 2>&1; tail -3 gpurun_out/r2f_pytest_conv.log
timeout 1500 python -m pytest tests/test_gpu_model.py -q -s --timeout=900 > gpurun_out/r2f_pytest_model.log 2>&1; tail -3 gpurun_out/r2f_pytest_model.log; grep -n "configs\[" gpurun_out/r2f_pytest_model.log
timeout 600 python bench.py --model CpnResNet18FPN --batch 8 --no-cpu-baseline --profile-layers > gpurun_out/r2f_bench_cfg1.json 2> gpurun_out/r2f_cfg1_layers.txt; cut -c1-250 gpurun_out/r2f_bench_cfg1.json; grep -n "bilinear\|refinement\|total" gpurun_out/r2f_cfg1_layers.txt | tail -5
timeout 600 python bench.py --model CpnResNet50FPN --batch 8 --tile 1024 --precision fp8 --no-cpu-baseline --steps 10 > gpurun_out/r2f_bench_cfg4.json 2> gpurun_out/r2f_cfg4.err; cut -c1-250 gpurun_out/r2f_bench_cfg4.json; tail -2 gpurun_out/r2f_cfg4.err
timeout 600 python bench.py --model CpnResNet50FPN --batch 8 --tile 1024 --no-cpu-baseline --steps 10 > gpurun_out/r2f_bench_cfg4_bf16.json 2> gpurun_out/r2f_cfg4b.err; cut -c1-250 gpurun_out/r2f_bench_cfg4_bf16.json
