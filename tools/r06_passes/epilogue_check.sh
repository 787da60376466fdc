#!/bin/bash
# Round 6: fused-head epilogue with the bias / multiplier table in LDS: tests that pin the fused heads + workgroup phases before / after
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sparse_heads.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -3
D=$PWD/celldetection_amd/build/variants
CPN_HIP_LIB=$D/libcpn_clock1.so CPN_HIP_GRAPH=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep -E "CLK|PHASES" | tail -600 > gpurun_out/r06_phases_in_graph_after.txt
CPN_HIP_LIB=$D/libcpn_clock1f8.so CPN_HIP_GRAPH=0 python bench.py --model CpnResNet50FPN --batch 8 --tile 1024 --precision fp8 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep -E "CLK|PHASES" | tail -700 > gpurun_out/r06_phases_configs4_fp8_after.txt
for z in 0; do CPN_MB_ZERO=$z python tools/conv_microbench.py head7 2>&1 | grep -v amdgpu; CPN_MB_FP8=1 CPN_MB_ZERO=$z python tools/conv_microbench.py head7 k5 2>&1 | grep -v amdgpu; done
python bench.py --steps 20 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-160
python bench.py --model CpnResNet50FPN --batch 8 --tile 1024 --precision fp8 --no-cpu-baseline --no-extras --steps 10 2>/dev/null | cut -c1-160
