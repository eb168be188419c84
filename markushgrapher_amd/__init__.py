"""MI355X-native MarkushGrapher-2 forward path."""
import os as _os

# HIP runtime knob, read when libamdhip64 is loaded (i.e. at `import torch`): keep kernel arguments in device memory
# instead of host-coherent memory.  Every launch starts with scalar loads of its arguments; the decode step is a chain of
# ~150 latency-sized launches, and host-resident arguments cost it ~1-2 % (eager launches: 62 -> 72 images/s).
# Import this package before torch for it to take effect; a value already set by the caller is respected.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# Hardware queues the HIP runtime spreads its streams over (default 4, one of them taken by the null stream): with the batches
# in flight of markushgrapher_amd/inflight.py every execution context needs a queue of its own - two contexts sharing one run
# back to back (4 contexts: 102 images/s on 4 queues, 116 on 8 or 16; profiles/r03_inflight_ab.txt).  16 rather than 8: a process
# that also runs RCCL collectives needs queues for RCCL's streams too - with 8, the all-gather of the ids landed on a context's queue
# (115 -> 98 images/s with a one-rank group on one GPU, back to 115 with 16).  More than 4 busy queues
# oversubscribe the chip's compute pipes and collapse (5 contexts: 80 images/s), which is why InFlight caps at 4.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
