#!/bin/bash
# Round 6: GPU suite + bench line + shader clock of the conv kernels inside the flagship graph (CPN_EXP_CLOCK build, eager launches)
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r06_final_gpu_tests.txt
python bench.py > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err
CPN_HIP_LIB=$PWD/celldetection_amd/build/variants/libcpn_clock.so CPN_HIP_GRAPH=0 python bench.py --steps 2 --warmup 1 2>&1 | grep CLK | tail -400 > gpurun_out/r06_final_clock_in_graph.txt
