#!/bin/bash
# round 6: FETCH_SIZE calibration for the conv kernels' two access patterns (tools/probes/fetch_calibration.hip), and the request-size
# counters gfx950 exposes (TCC_EA0_RDREQ_128B / _64B / _32B): bytes = 128 * n128 + 64 * n64 + 32 * n32
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=/tmp/prof_fetchcal; mkdir -p $P
(timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/f -o b -- tools/probes/fetch_calibration.bin) > $P/f.log 2>&1
grep "bytes touched" $P/f.log
python tools/summarize_rocprof.py $P/f | grep -v "^==" | cut -c1-160
(timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $P/r -o b -- tools/probes/fetch_calibration.bin) > $P/r.log 2>&1
python tools/summarize_rocprof.py $P/r | grep -v "^==" | cut -c1-220
