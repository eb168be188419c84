"""ctypes loader for the HIP library.  The product path has NO fallback: if libmgrapher_hip.so is missing or
cannot be loaded, importing/using the engine fails loudly (build it with `python __graft_entry__.py` or
`python markushgrapher_amd/csrc/build.py hip`)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmgrapher_hip.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the MI355X HIP library is not built. There is no CPU fallback; run "
                "`python __graft_entry__.py` (or markushgrapher_amd/csrc/build.py hip) first.")
        # torch (the device-memory carrier) bundles its own libamdhip64; import it FIRST so this library binds to the
        # same HIP runtime instance — two runtimes in one process do not share streams or allocations
        # (symptom: hipErrorNoDevice from the first launch).
        import torch  # noqa: F401
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib
