#!/bin/bash
# MFMA-busy counters of a short bench run (counters in their own rocprofv3 run with --kernel-trace only) -> gpurun_out/<tag>_pmc_mfma.md
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmcm_$tag
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/pmcm_$tag -o $tag -- python $root/bench.py --steps 1 --warmup 0 --new-tokens 2 --no-cpu-baseline --no-extra-runs --no-pmc "$@" > $root/gpurun_out/${tag}_pmc_mfma.log 2>&1 || true
db=$(find /tmp/pmcm_$tag -name "*.db" | head -1)
python $root/tools/pmc_mfma_table.py $db $root/gpurun_out/${tag}_pmc_mfma.md
