#!/bin/bash
# rocprofv3 kernel trace of the ChemicalOCR stage's vision tower + prefill (32 pages, a few decode steps) -> gpurun_out/r05_s_ocr_prefill_kernel_stats.md
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_ocrpre
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_ocrpre -o ocrpre -- python $root/tools/ocr_prefill_probe.py > $root/gpurun_out/r05_s_ocr_prefill_profiled.txt 2>&1 || true
db=$(find /tmp/prof_ocrpre -name "*.db" | head -1)
python $root/tools/rocpd_stats.py $db $root/gpurun_out/r05_s_ocr_prefill_kernel_stats.md --by-grid > /dev/null
head -45 $root/gpurun_out/r05_s_ocr_prefill_kernel_stats.md
