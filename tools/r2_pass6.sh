#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python bench.py --model CpnResNet50FPN --batch 8 --tile 1024 --precision fp8 --no-cpu-baseline --steps 3 --warmup 2 --profile-layers > gpurun_out/r2g_bench_cfg4.json 2> gpurun_out/r2g_cfg4_layers.txt; cut -c1-200 gpurun_out/r2g_bench_cfg4.json; grep -v "^  *[0-9]* conv k1" gpurun_out/r2g_cfg4_layers.txt | tail -30
timeout 600 python bench.py --precision fp8 --no-cpu-baseline --steps 10 > gpurun_out/r2g_bench_fp8.json 2>/dev/null; cut -c1-200 gpurun_out/r2g_bench_fp8.json
