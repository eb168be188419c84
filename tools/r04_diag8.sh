#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_diag8b.txt
: > $out
for cfg in "0 3 2" "1 3 2" "1 3 0" "0 4 2" "1 4 2" "1 3 1"; do
  set -- $cfg
  MG_ENC_SPLIT=$1 MG_GEMM_VARIANT=$2 MG_ENC_SPLIT_CUS=$3 timeout 300 python tools/att_bench.py 2>&1 | tail -1 | sed "s/^/split $1 gemm variant $2 cus-div $3: /" >> $out
done
timeout 300 python tools/enc_inflight_probe.py --reps 4 2>&1 | grep encoder >> $out
cat $out
