#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_diag6.txt
: > $out
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "attention_encoder" 2>&1 | tail -3 >> $out
for v in 0 1 0 1; do MG_ATT_VARIANT=$v timeout 300 python tools/att_bench.py 2>&1 | tail -1 | sed "s/^/att variant $v: /" >> $out; done
cat $out
