#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_diag3.txt
: > $out
for x in 0 1 2 3 4 8 9 11 15; do MG_PP_EXP=$x timeout 300 python tools/kbench.py ppexp none 2>&1 | grep "pp exp" >> $out; done
cat $out
