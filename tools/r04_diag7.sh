#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_diag7.txt
: > $out
for v in 0 1 0 1 0 1; do MG_PP_BALANCE=$v timeout 300 python tools/att_bench.py 2>&1 | tail -1 | sed "s/^/pp balance $v: /" >> $out; done
cat $out
