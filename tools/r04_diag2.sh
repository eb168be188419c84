#!/bin/bash
# round-4 call 2: ping-pong GEMM: parity tests on the GPU, isolated timing against the two-stage kernel, whole encoder per variant
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_diag2.txt
: > $out
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "tile_kernel or row_tile_list or encoder_deferred" >> $out 2>&1
timeout 600 python tools/kbench.py encgemm none >> $out 2>&1
for v in 3 5 6; do MG_GEMM_VARIANT=$v timeout 300 python tools/att_bench.py 2>&1 | tail -1 | sed "s/^/variant $v: /" >> $out; done
cat $out
