#!/bin/bash
# Round 6: where a conv workgroup's time goes inside the flagship graph -- set-up / prologue + main loop / epilogue (incl. the fused
# ReadOut tails) of one workgroup per launch (tools/build_variant.sh clock1 -DCPN_EXP_CLOCK=1; eager launches; random operands)
cd "$GRAFT_REPO_ROOT"

CPN_HIP_LIB=$PWD/celldetection_amd/build/variants/libcpn_clock1.so CPN_HIP_GRAPH=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep -E "CLK|PHASES" | tail -600 > gpurun_out/r06_phases_in_graph.txt
python tools/clock_probe.py 20 > gpurun_out/r06_clock_probe_graph.json 2>/dev/null; cat gpurun_out/r06_clock_probe_graph.json
