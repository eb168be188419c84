#!/bin/bash
# A/B of the whole encoder: current library against tools/_build/libmgrapher_prev.so (built from the previous commit), same box
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_ab_prev.txt
: > $out
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "attention" 2>&1 | grep -E "passed|failed" >> $out
for i in 1 2 3; do
  MG_LIB_PATH=tools/_build/libmgrapher_prev.so timeout 300 python tools/att_bench.py 2>&1 | tail -1 | sed "s/^/prev: /" >> $out
  timeout 300 python tools/att_bench.py 2>&1 | tail -1 | sed "s/^/new:  /" >> $out
done
cat $out
