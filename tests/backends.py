"""Two ways to run the same C-ABI calls in the tests:

  hip : libmgrapher_hip.so on a real MI355X, device buffers carried by torch tensors  (tests marked gpu)
  emu : the SAME kernel sources compiled for the CPU SIMT emulator (tools/simt_emu) — test infrastructure that
        checks the kernels' index math / layouts in the GPU-less build container.  Not a product path.
"""
import ctypes
import os

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


class Buf:
    def __init__(self, be, arr):
        self.be = be
        self.dtype = arr.dtype
        self.shape = arr.shape
        if be.name == "emu":
            self.a = np.ascontiguousarray(arr).copy()
            self.ptr = self.a.ctypes.data
        else:
            import torch
            src = np.ascontiguousarray(arr)
            if src.dtype == np.uint16:
                t = torch.from_numpy(src.view(np.int16).copy())
            else:
                t = torch.from_numpy(src.copy())
            self.a = t.cuda()
            self.ptr = self.a.data_ptr()

    def numpy(self):
        if self.be.name == "emu":
            return self.a
        self.be.sync()
        out = self.a.cpu().numpy()
        return out.view(np.uint16) if self.dtype == np.uint16 else out


class Backend:
    def __init__(self, name):
        self.name = name
        self._keep = []
        if name == "emu":
            import importlib.util
            spec = importlib.util.spec_from_file_location(
                "mg_build", os.path.join(ROOT, "markushgrapher_amd", "csrc", "build.py"))
            b = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(b)
            self.lib = ctypes.CDLL(b.build_emu())
            self.stream = None
        else:
            import torch
            assert torch.cuda.is_available(), "hip backend needs a GPU"
            from markushgrapher_amd import _lib
            self.lib = _lib.load()
            self.torch = torch
            self.stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def buf(self, arr):
        b = Buf(self, np.asarray(arr))
        self._keep.append(b)      # buffers passed inline as be.p(be.buf(..)) must outlive the call
        return b

    def zeros(self, shape, dtype):
        return self.buf(np.zeros(shape, dtype))

    def sync(self):
        if self.name != "emu":
            self.torch.cuda.synchronize()

    def p(self, b):
        if b is None:
            return ctypes.c_void_p(0)
        return ctypes.c_void_p(b.ptr)


_cache = {}


def get_backend(name):
    if name not in _cache:
        _cache[name] = Backend(name)
    _cache[name].sync()
    _cache[name]._keep.clear()    # a test calls get_backend() first: drop the previous test's buffers
    return _cache[name]


class NumpyMem:
    """Memory provider for markushgrapher_amd.engine.Engine on the emulator backend (host memory)."""

    def empty(self, shape, dtype):
        return np.zeros(shape, dtype)

    zeros = empty

    def asarray(self, x, dtype):
        if not isinstance(x, np.ndarray):
            x = x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
        return np.ascontiguousarray(x).astype(dtype, copy=True)

    def copy(self, h):
        return np.array(h, copy=True)

    def ptr(self, h):
        return ctypes.c_void_p(h.ctypes.data)

    def stream(self):
        return ctypes.c_void_p(0)

    def sync(self):
        pass

    def numpy(self, h):
        return h


def make_engine(be_name, shape, sd, max_decode_len=64):
    """Engine over the chosen backend with the given fp32 (bf16-exact) state dict loaded."""
    from markushgrapher_amd.engine import Engine
    be = get_backend(be_name)
    if be_name == "emu":
        eng = Engine(shape, lib=be.lib, mem=NumpyMem(), max_decode_len=max_decode_len)
    else:
        eng = Engine(shape, max_decode_len=max_decode_len)
    if sd is not None:
        eng.load_state_dict(sd)
    return eng
