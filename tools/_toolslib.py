"""Loader of the TOOLS build of the library (tools/_build/libmgrapher_tools.so = the product sources + -DMG_TOOLS: phase-stamped
trace kernels, what-if GEMM variants).  Profiling scripts only; nothing in markushgrapher_amd/ can load it."""
import ctypes
import importlib.util
import os

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def load():
    spec = importlib.util.spec_from_file_location("mg_build", os.path.join(ROOT, "markushgrapher_amd", "csrc", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    so = b.TOOLS_SO if os.path.exists(b.TOOLS_SO) else b.build_tools()
    import torch  # noqa: F401  (same HIP runtime instance as the tensors, see markushgrapher_amd/_lib.py)
    return ctypes.CDLL(so)
