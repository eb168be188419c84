"""MI355X-native MarkushGrapher-2 forward path."""
import os as _os

# HIP runtime knob, read when libamdhip64 is loaded (i.e. at `import torch`): keep kernel arguments in device memory
# instead of host-coherent memory.  Every launch starts with scalar loads of its arguments; the decode step is a chain of
# ~150 latency-sized launches, and host-resident arguments cost it ~1-2 % (eager launches: 62 -> 72 images/s).
# Import this package before torch for it to take effect; a value already set by the caller is respected.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
