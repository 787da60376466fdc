#!/bin/bash
# Round-end GPU pass (run on the MI355X box through gpurun, from the repo root):
#   tools/gpu_round_pass.sh <tag> [tests] [bench] [profile] [fp8] [slide] [configs] [sparse]
# writes everything under gpurun_out/ (the summaries that should be judged are then copied into profiles/).
TAG=${1:-r02}; shift
WHAT=${*:-tests bench profile}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
P=/tmp/prof_$TAG; mkdir -p $P
G="python tools/run_graph_only.py 5"
for w in $WHAT; do case $w in
tests)
  timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log;;
bench)
  timeout 600 python bench.py --profile-layers > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_per_layer_timing.txt; cat gpurun_out/${TAG}_bench_n1.json;;
profile)
  (timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o b -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline) > $P/trace.log 2>&1
  (timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gtrace -o b -- $G) > $P/gtrace.log 2>&1
  (timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -o b -- $G) > $P/fetch.log 2>&1
  (timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -o b -- $G) > $P/write.log 2>&1
  (timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY --output-format csv -d $P/mfma -o b -- $G) > $P/mfma.log 2>&1
  (echo "### bench.py --steps 5 --warmup 2 under rocprofv3 --kernel-trace --stats"
   python tools/summarize_rocprof.py $P/trace; echo
   echo "### tools/run_graph_only.py 5 (5 conv-graph executions, batch 16): kernel-trace, FETCH_SIZE, WRITE_SIZE, MFMA/SQ counters"
   python tools/summarize_rocprof.py $P/gtrace $P/fetch $P/write $P/mfma) > gpurun_out/${TAG}_rocprofv3_summary.txt 2>&1
  python tools/summarize_rocprof.py --traffic-json $P/fetch $P/write 5 CpnResNeXt101UNet/b16/t512/bf16 gpurun_out/${TAG}_traffic.json \
    "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of tools/run_graph_only.py 5 (${TAG}); see profiles/${TAG}_rocprofv3_summary.txt"
  grep -h "^{" $P/trace.log > gpurun_out/${TAG}_bench_lines_under_rocprof.txt
  grep -A8 "run_graph_only" gpurun_out/${TAG}_rocprofv3_summary.txt | cut -c1-150;;
slide)
  timeout 600 python bench.py --workload slide > gpurun_out/${TAG}_bench_slide_n1.json 2> gpurun_out/${TAG}_bench_slide.err; cut -c1-900 gpurun_out/${TAG}_bench_slide_n1.json
  CPN_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload slide --slide 4096 > gpurun_out/${TAG}_bench_slide_rccl1.json 2> gpurun_out/${TAG}_bench_slide_rccl1.err; cut -c1-300 gpurun_out/${TAG}_bench_slide_rccl1.json;;
configs)
  timeout 600 python bench.py --model CpnResNet18FPN --batch 8 --no-cpu-baseline --profile-layers > gpurun_out/${TAG}_bench_cfg1.json 2> gpurun_out/${TAG}_cfg1_layers.txt; cut -c1-200 gpurun_out/${TAG}_bench_cfg1.json
  timeout 600 python bench.py --model CpnResNet50FPN --batch 8 --tile 1024 --precision fp8 --no-cpu-baseline --steps 10 > gpurun_out/${TAG}_bench_cfg4_fp8.json 2> gpurun_out/${TAG}_cfg4.err; cut -c1-200 gpurun_out/${TAG}_bench_cfg4_fp8.json
  timeout 600 python bench.py --model CpnResNet50FPN --batch 4 --tile 1024 --precision fp8 --no-cpu-baseline --steps 10 --profile-layers > gpurun_out/${TAG}_bench_cfg4_fp8_b4.json 2> gpurun_out/${TAG}_cfg4_b4_layers.txt; cut -c1-200 gpurun_out/${TAG}_bench_cfg4_fp8_b4.json; tail -4 gpurun_out/${TAG}_cfg4_b4_layers.txt;;
sparse)  # score-gated heads (experimental): acceptance tests, then the bench line with and without them on the same box
  CPN_TEST_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_gpu_sparse_heads.py -q > gpurun_out/${TAG}_pytest_sparse.log 2>&1; tail -5 gpurun_out/${TAG}_pytest_sparse.log
  timeout 300 python tools/sparse_microbench.py > gpurun_out/${TAG}_sparse_microbench.txt 2>&1; cat gpurun_out/${TAG}_sparse_microbench.txt
  timeout 600 python bench.py --no-cpu-baseline --sparse-heads > gpurun_out/${TAG}_bench_sparse_n1.json 2> gpurun_out/${TAG}_bench_sparse.err; cut -c1-400 gpurun_out/${TAG}_bench_sparse_n1.json
  timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_dense_n1.json 2> /dev/null; cut -c1-200 gpurun_out/${TAG}_bench_dense_n1.json;;
fp8)
  timeout 600 python bench.py --precision fp8 --no-cpu-baseline > gpurun_out/${TAG}_bench_fp8_n1.json 2> gpurun_out/${TAG}_bench_fp8.err; cat gpurun_out/${TAG}_bench_fp8_n1.json | cut -c1-300
  (timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/f8trace -o b -- python bench.py --precision fp8 --steps 5 --warmup 2 --no-cpu-baseline) > $P/f8trace.log 2>&1
  (echo "### bench.py --precision fp8 --steps 5 --warmup 2 under rocprofv3 --kernel-trace --stats"; python tools/summarize_rocprof.py $P/f8trace) > gpurun_out/${TAG}_fp8_rocprofv3_summary.txt 2>&1
  head -12 gpurun_out/${TAG}_fp8_rocprofv3_summary.txt | cut -c1-150;;
esac; done
