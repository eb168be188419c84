#!/bin/bash
# rocprofv3 kernel trace of the beam queue (mg_generate_stream_beam, 32 image slots x 5 rows, beam-5 + EOS) -> gpurun_out/r05_beamq_kernel_stats.md
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_beamq
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_beamq -o beamq -- python $root/tools/beam_queue_probe.py --queue 2 --slots 32 > $root/gpurun_out/r05_beamq_profiled.txt 2>&1 || true
db=$(find /tmp/prof_beamq -name "*.db" | head -1)
python $root/tools/rocpd_stats.py $db $root/gpurun_out/r05_beamq_kernel_stats.md --by-grid > /dev/null
head -45 $root/gpurun_out/r05_beamq_kernel_stats.md
