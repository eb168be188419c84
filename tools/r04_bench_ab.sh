#!/bin/bash
# headline bench A/B over encoder GEMM variants (MG_GEMM_VARIANT), 4 batches in flight and one
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_bench_ab.txt
: > $out
for v in "$@"; do
  for inf in 4 1; do
    MG_GEMM_VARIANT=$v timeout 600 python bench.py --steps 12 --warmup 2 --no-extra-runs --no-cpu-baseline --no-pmc --inflight $inf 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); p=d.get('phases',{})
        print('variant $v inflight $inf: %.2f images/s  %.1f ms/step  enc %.2f ms  enc_mfma %.4f  step %.4f ms' % (d['value'], d['ms_per_step'], p.get('encoder_ms',0), p.get('enc_mfma_frac',0), p.get('decode_step_ms',0)))
" >> $out
  done
done
cat $out
