#!/bin/bash
# Round-end GPU pass (run on the MI355X box through gpurun, from the repo root):
#   tools/gpu_round_pass.sh <tag> [tests] [bench] [profile] [sparse] [slide] [configs] [c4traffic] [fp8]
# writes everything under gpurun_out/ (the summaries that should be judged are then copied into profiles/).
TAG=${1:-r04}; shift
WHAT=${*:-tests bench profile}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
P=/tmp/prof_$TAG; mkdir -p $P
S="python tools/summarize_rocprof.py"
pmc_traffic() {  # <name> <key> <graph-only args...>: FETCH_SIZE / WRITE_SIZE passes (separate runs) -> traffic.json entry
  local name=$1 key=$2; shift 2
  (timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/${name}_fetch -o b -- python tools/run_graph_only.py 5 "$@") > $P/${name}_fetch.log 2>&1
  (timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/${name}_write -o b -- python tools/run_graph_only.py 5 "$@") > $P/${name}_write.log 2>&1
  $S --traffic-json $P/${name}_fetch $P/${name}_write 5 $key gpurun_out/${TAG}_traffic.json \
    "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of tools/run_graph_only.py 5 $* (${TAG})"
}
for w in $WHAT; do case $w in
tests)
  timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log;;
bench)
  timeout 600 python bench.py --profile-layers > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_per_layer_timing.txt; cat gpurun_out/${TAG}_bench_n1.json | cut -c1-400;;
profile)
  (timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o b -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras) > $P/trace.log 2>&1
  (timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gtrace -o b -- python tools/run_graph_only.py 5) > $P/gtrace.log 2>&1
  pmc_traffic main CpnResNeXt101UNet/b16/t512/bf16
  (timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY --output-format csv -d $P/mfma -o b -- python tools/run_graph_only.py 5) > $P/mfma.log 2>&1
  (echo "### bench.py --steps 5 --warmup 2 under rocprofv3 --kernel-trace --stats (hipGraph replay of the conv graph)"
   $S $P/trace; echo
   echo "### tools/run_graph_only.py 5 (5 eager conv-graph executions, batch 16): kernel-trace, FETCH_SIZE, WRITE_SIZE, MFMA/SQ counters"
   $S $P/gtrace $P/main_fetch $P/main_write $P/mfma) > gpurun_out/${TAG}_rocprofv3_summary.txt 2>&1
  grep -h "^{" $P/trace.log > gpurun_out/${TAG}_bench_lines_under_rocprof.txt
  grep -A8 "run_graph_only" gpurun_out/${TAG}_rocprofv3_summary.txt | cut -c1-150;;
sparse)  # score-gated heads: density sweep + the gated bench line (second line; the default line stays the dense graph)
  timeout 300 python tools/sparse_microbench.py > gpurun_out/${TAG}_sparse_density_sweep.txt 2>&1; cat gpurun_out/${TAG}_sparse_density_sweep.txt
  timeout 600 python bench.py --no-cpu-baseline --sparse-heads > gpurun_out/${TAG}_bench_sparse_heads_n1.json 2> gpurun_out/${TAG}_bench_sparse.err; cut -c1-300 gpurun_out/${TAG}_bench_sparse_heads_n1.json
  (timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/sptrace -o b -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sparse-heads) > $P/sptrace.log 2>&1
  (echo "### bench.py --sparse-heads --steps 5 --warmup 2 under rocprofv3 --kernel-trace --stats"; $S $P/sptrace) > gpurun_out/${TAG}_sparse_heads_rocprofv3_summary.txt 2>&1
  grep -E "sparse_heads_kernel|kernel  " gpurun_out/${TAG}_sparse_heads_rocprofv3_summary.txt | cut -c1-150;;
slide)
  timeout 600 python bench.py --workload slide > gpurun_out/${TAG}_bench_slide_n1.json 2> gpurun_out/${TAG}_bench_slide.err; cut -c1-900 gpurun_out/${TAG}_bench_slide_n1.json
  CPN_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload slide --slide 4096 > gpurun_out/${TAG}_bench_slide_4096_rccl_1rank.json 2> gpurun_out/${TAG}_bench_slide_rccl1.err; cut -c1-300 gpurun_out/${TAG}_bench_slide_4096_rccl_1rank.json;;
configs)
  timeout 600 python bench.py --model CpnResNet18FPN --batch 8 --no-cpu-baseline --profile-layers > gpurun_out/${TAG}_bench_configs1_resnet18fpn.json 2> gpurun_out/${TAG}_configs1_per_layer_timing.txt; cut -c1-200 gpurun_out/${TAG}_bench_configs1_resnet18fpn.json
  (timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/c1trace -o b -- python tools/run_graph_only.py 5 CpnResNet18FPN 8 512 bf16) > $P/c1trace.log 2>&1
  pmc_traffic c1 CpnResNet18FPN/b8/t512/bf16 CpnResNet18FPN 8 512 bf16
  (echo "### tools/run_graph_only.py 5 CpnResNet18FPN 8 512 bf16 (BASELINE configs[1]): kernel-trace, FETCH_SIZE, WRITE_SIZE"; $S $P/c1trace $P/c1_fetch $P/c1_write) > gpurun_out/${TAG}_configs1_rocprofv3_summary.txt 2>&1
  timeout 600 python bench.py --model CpnResNet50FPN --batch 8 --tile 1024 --no-cpu-baseline --steps 10 > gpurun_out/${TAG}_bench_configs4_resnet50fpn_bf16.json 2> gpurun_out/${TAG}_cfg4_bf16.err; cut -c1-200 gpurun_out/${TAG}_bench_configs4_resnet50fpn_bf16.json
  timeout 600 python bench.py --model CpnResNet50FPN --batch 8 --tile 1024 --precision fp8 --no-cpu-baseline --steps 10 > gpurun_out/${TAG}_bench_configs4_resnet50fpn_fp8.json 2> gpurun_out/${TAG}_cfg4.err; cut -c1-200 gpurun_out/${TAG}_bench_configs4_resnet50fpn_fp8.json
  (timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/c4trace -o b -- python tools/run_graph_only.py 3 CpnResNet50FPN 4 1024 fp8) > $P/c4trace.log 2>&1
  (echo "### tools/run_graph_only.py 3 CpnResNet50FPN 4 1024 fp8 (BASELINE configs[4], the engine's sub-batch of 4 tiles): kernel-trace"; $S $P/c4trace) > gpurun_out/${TAG}_configs4_fp8_rocprofv3_summary.txt 2>&1;;
c4traffic)  # HBM traffic of BASELINE configs[4] (batch 8 x 1024^2 = two engine sub-batches per graph execution), bf16 and fp8
  pmc_traffic c4b CpnResNet50FPN/b8/t1024/bf16 CpnResNet50FPN 8 1024 bf16
  pmc_traffic c4f CpnResNet50FPN/b8/t1024/fp8 CpnResNet50FPN 8 1024 fp8;;
fp8)
  timeout 600 python bench.py --precision fp8 --no-cpu-baseline > gpurun_out/${TAG}_bench_fp8_n1.json 2> gpurun_out/${TAG}_bench_fp8.err; cat gpurun_out/${TAG}_bench_fp8_n1.json | cut -c1-300
  pmc_traffic f8 CpnResNeXt101UNet/b16/t512/fp8 CpnResNeXt101UNet 16 512 fp8
  (timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/f8trace -o b -- python tools/run_graph_only.py 5 CpnResNeXt101UNet 16 512 fp8) > $P/f8trace.log 2>&1
  (echo "### tools/run_graph_only.py 5 CpnResNeXt101UNet 16 512 fp8 under rocprofv3: kernel-trace, FETCH_SIZE, WRITE_SIZE"; $S $P/f8trace $P/f8_fetch $P/f8_write) > gpurun_out/${TAG}_fp8_rocprofv3_summary.txt 2>&1
  head -12 gpurun_out/${TAG}_fp8_rocprofv3_summary.txt | cut -c1-150;;
esac; done
