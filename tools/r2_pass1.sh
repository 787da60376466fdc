#!/bin/bash
# round-2 GPU pass 1: full GPU test suite, conv tests on the halo-spreading variant, DMA ablations, bench lines
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x --timeout=900 > gpurun_out/r2a_pytest_gpu.log 2>&1; tail -5 gpurun_out/r2a_pytest_gpu.log
CPN_HIP_LIB=$PWD/celldetection_amd/build/variants/libcpn_spread.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "test_conv" > gpurun_out/r2a_pytest_spread.log 2>&1; tail -2 gpurun_out/r2a_pytest_spread.log
timeout 600 tools/ab.sh "head7 dec3b dec3 dec3cat c64 ref7 pw1024 grp" base spread nowdma nohdma nodma nomfma > gpurun_out/r2a_ab.txt 2>&1; cat gpurun_out/r2a_ab.txt
timeout 600 python bench.py --profile-layers > gpurun_out/r2a_bench_n1.json 2> gpurun_out/r2a_per_layer_timing.txt; cut -c1-600 gpurun_out/r2a_bench_n1.json
timeout 600 python bench.py --workload slide > gpurun_out/r2a_bench_slide_n1.json 2> gpurun_out/r2a_bench_slide.err; cat gpurun_out/r2a_bench_slide_n1.json; tail -3 gpurun_out/r2a_bench_slide.err
