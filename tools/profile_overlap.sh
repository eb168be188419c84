#!/bin/bash
# kernel trace of a short in-flight bench run -> per-kernel table (tools/rocpd_stats.py) + decode kernels beside / not beside encoder kernels (tools/rocpd_overlap.py)
# usage: tools/profile_overlap.sh <tag> [bench args]
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_$tag
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_$tag -o $tag -- python $root/bench.py --steps 16 --warmup 1 --no-extra-runs --no-cpu-baseline --no-pmc "$@" > $root/gpurun_out/${tag}_profiled_bench.json 2> $root/gpurun_out/${tag}_profiled_bench.err || true
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python $root/tools/rocpd_stats.py $db $root/gpurun_out/${tag}_kernel_stats.md --by-grid > /dev/null
python $root/tools/rocpd_overlap.py $db > $root/gpurun_out/${tag}_overlap.md
cat $root/gpurun_out/${tag}_overlap.md
