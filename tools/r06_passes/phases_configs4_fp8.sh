cd "$GRAFT_REPO_ROOT"
CPN_HIP_LIB=$PWD/celldetection_amd/build/variants/libcpn_clock1f8.so CPN_HIP_GRAPH=0 python bench.py --model CpnResNet50FPN --batch 8 --tile 1024 --precision fp8 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep -E "CLK|PHASES" | tail -700 > gpurun_out/r06_phases_configs4_fp8.txt
wc -l gpurun_out/r06_phases_configs4_fp8.txt
