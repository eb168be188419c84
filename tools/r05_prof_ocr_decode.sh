#!/bin/bash
# rocprofv3 kernel trace of the ChemicalOCR stage at the configs[4] workload: ONE context, 128 pages through 128 decode rows (1176 steps,
# twice: warm-up + timed) -> gpurun_out/r05_m_ocr_decode_kernel_stats.md
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_ocrdec
timeout 1200 rocprofv3 --kernel-trace -d /tmp/prof_ocrdec -o ocrdec -- python $root/tools/ocr_rows_probe.py 1:128:128 > $root/gpurun_out/r05_m_ocr_decode_profiled.txt 2>&1 || true
db=$(find /tmp/prof_ocrdec -name "*.db" | head -1)
python $root/tools/rocpd_stats.py $db $root/gpurun_out/r05_m_ocr_decode_kernel_stats.md --by-grid > /dev/null
head -40 $root/gpurun_out/r05_m_ocr_decode_kernel_stats.md
