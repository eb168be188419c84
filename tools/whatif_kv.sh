#!/bin/bash
# what-if runs (tools build, WRONG results): the QKV projection's K/V cache append with 4 contexts x 128 rows in flight
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $root/gpurun_out
cd /tmp
for wi in 0 1 2; do
  rm -rf /tmp/prof_wi$wi
  MG_WHATIF_KV=$wi timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_wi$wi -o wi$wi -- python $root/tools/inflight_probe.py --inflight 4 --batch 128 --clones --batches 1 --only --tools-lib --new-tokens 128 2>&1 | grep "in flight"
  db=$(find /tmp/prof_wi$wi -name "*.db" | head -1)
  python $root/tools/rocpd_stats.py $db $root/gpurun_out/whatif_kv_$wi.md --by-grid | grep "grid 2048x\|grid 192x512\|grid 128x512\|grid 64x1024\|grid 1038" | grep -v "lds 3289\|lds 16512" | head -8
done
