#!/bin/bash
# round-4 diagnostics, call 1: attention what-ifs, encoder contexts in flight, encoder GEMM baseline
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_diag1.txt
: > $out
python tools/att_bench.py >> $out 2>&1
for x in 1 2 3 4 7 8 9 15; do MG_ATT_EXP=$x timeout 300 python tools/att_bench.py 2>&1 | tail -1 >> $out; done
timeout 600 python tools/enc_inflight_probe.py >> $out 2>&1
timeout 600 python tools/kbench.py encgemm none >> $out 2>&1
cat $out
