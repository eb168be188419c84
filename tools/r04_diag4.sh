#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_diag4.txt
: > $out
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "tile_kernel or row_tile_list or encoder_deferred" 2>&1 | tail -3 >> $out
timeout 600 python tools/kbench.py encgemm none >> $out 2>&1
for x in 1 2 3 8; do MG_PP_EXP=$x timeout 300 python tools/kbench.py ppexp none 2>&1 | grep "pp exp" | grep "variant [78]" >> $out; done
for v in 3 6 7 8; do MG_GEMM_VARIANT=$v timeout 300 python tools/att_bench.py 2>&1 | tail -1 | sed "s/^/variant $v: /" >> $out; done
cat $out
