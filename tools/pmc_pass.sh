#!/bin/bash
# One rocprofv3 --pmc pass over a short bench run (counters in their own run with --kernel-trace only, as gpurun requires).
# usage: tools/pmc_pass.sh <tag> "<counters>" [bench args]      -> gpurun_out/<tag>_pmc.md
set -e
tag=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_$tag
timeout 600 rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/pmc_$tag -o $tag -- python $root/bench.py --steps 1 --warmup 0 --new-tokens 2 --no-cpu-baseline "$@" > $root/gpurun_out/${tag}_pmc.log 2>&1 || true
db=$(find /tmp/pmc_$tag -name "*.db" | head -1)
python $root/tools/rocpd_pmc.py $db $root/gpurun_out/${tag}_pmc.md > /dev/null
