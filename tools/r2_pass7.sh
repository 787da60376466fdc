#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python tools/conv_microbench.py bl7 bl7s head7 2>&1 | grep -v amdgpu.ids
CPN_MB_FP8=1 python tools/conv_microbench.py bl7 bl7s head7 2>&1 | grep -v amdgpu.ids
P=/tmp/prof_bl; mkdir -p $P
(CPN_MB_FP8=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $P/a -o b -- python tools/conv_microbench.py bl7s head7) > $P/a.log 2>&1
(CPN_MB_FP8=1 timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/b -o b -- python tools/conv_microbench.py bl7s head7) > $P/b.log 2>&1
python tools/summarize_rocprof.py $P/a $P/b > gpurun_out/r2h_bl_fp8_pmc.txt 2>&1; cut -c1-400 gpurun_out/r2h_bl_fp8_pmc.txt | grep -v "^==" | head -30
