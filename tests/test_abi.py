"""CPU: the C-ABI shared library loads (no GPU needed to dlopen it) and exports every symbol include/mgrapher.h
declares; argument validation that needs no device work fails with the documented error codes."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mgrapher.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mgk?_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_by_hip_library():
    from markushgrapher_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import importlib.util
        spec = importlib.util.spec_from_file_location("mg_build", os.path.join(ROOT, "markushgrapher_amd", "csrc", "build.py"))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        b.build_hip()
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from markushgrapher_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_create_validates_config_without_device_work():
    from markushgrapher_amd import _lib
    from markushgrapher_amd.engine import MgConfig
    lib = _lib.load()
    lib.mg_last_error.restype = C.c_char_p
    bad = MgConfig(500, 64, 16, 128, 2, 2, 4, 32, 128, 128, 64, 16, 3, 0, 1, 0, 1e-6, 64)   # d_kv = 16 unsupported
    model = C.c_void_p()
    assert lib.mg_create(C.byref(bad), C.byref(model)) == -5
    assert b"d_kv" in lib.mg_last_error()
    ok = MgConfig(500, 64, 64, 128, 2, 2, 2, 32, 128, 128, 64, 16, 3, 0, 1, 0, 1e-6, 64)
    assert lib.mg_create(C.byref(ok), C.byref(model)) == 0
    lib.mg_weights_bytes.restype = C.c_size_t
    lib.mg_weights_bytes.argtypes = [C.c_void_p]
    assert lib.mg_weights_bytes(model) > 100000
    need = C.c_size_t()
    lib.mg_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
    assert lib.mg_workspace_bytes(model, 2, 8, 5, 16, 12, 0, C.byref(need)) == 0 and need.value > 0
    base = need.value
    assert lib.mg_workspace_bytes(model, 2, 8, 5, 16, 12, 144, C.byref(need)) == 0 and need.value > base     # + e1 tokens
    # loading a tensor before binding an arena is a state error, an unknown key a key error (no device work)
    shp = (C.c_int64 * 1)(64)
    lib.mg_load_tensor.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]
    assert lib.mg_load_tensor(model, None, b"encoder.final_layer_norm.weight", C.c_void_p(16), 0, shp, 1) == -3
    lib.mg_destroy.argtypes = [C.c_void_p]
    lib.mg_destroy(model)


def test_cross_absorb_setting_and_workspace_rule():
    """mg_set_cross_absorb without device work: the default (2) picks the cross-attention form by the call's decode rows - from 96 rows on the
    weight-absorbed form, whose workspace holds ONE buffer of encoder states instead of per-layer K / V - 1 / 0 pin a form, < 0 queries, beam
    search always keeps K / V; bad arguments are refused."""
    from markushgrapher_amd import _lib
    from markushgrapher_amd.engine import MgConfig
    lib = _lib.load()
    lib.mg_last_error.restype = C.c_char_p
    cfg = MgConfig(500, 64, 64, 128, 2, 2, 2, 32, 128, 128, 64, 16, 3, 0, 1, 0, 1e-6, 64)
    model = C.c_void_p()
    assert lib.mg_create(C.byref(cfg), C.byref(model)) == 0
    lib.mg_set_cross_absorb.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.mg_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
    need = C.c_size_t()

    def ws(B, beams=1):
        assert lib.mg_workspace_bytes(model, B, 8, beams, 16, 0, 0, C.byref(need)) == 0
        return need.value
    assert lib.mg_set_cross_absorb(model, -1, 0) == 2                   # default: by the call's rows
    auto95, auto96, auto_beam = ws(95), ws(96), ws(32, 5)
    assert lib.mg_set_cross_absorb(model, 0, 0) == 2                    # K / V form always; returns the previous setting
    kv95, kv96, kv_beam = ws(95), ws(96), ws(32, 5)
    assert lib.mg_set_cross_absorb(model, 1, 2) == 0                    # absorbed for every greedy call, two key splits
    ab95, ab96, ab_beam = ws(95), ws(96), ws(32, 5)
    assert auto95 == kv95 and auto96 < kv96 and ab95 < kv95             # the rule; the absorbed form needs less
    assert auto_beam == kv_beam == ab_beam                              # beam search keeps the K / V form
    assert lib.mg_set_cross_absorb(model, -1, 0) == 1
    assert lib.mg_set_cross_absorb(model, 3, 0) < 0 and b"absorb" in lib.mg_last_error()
    assert lib.mg_set_cross_absorb(model, 1, 5) < 0
    assert lib.mg_set_cross_absorb(None, 1, 0) < 0
    lib.mg_destroy.argtypes = [C.c_void_p]
    lib.mg_destroy(model)
