"""OCSR vision branch "e1" on the MI355X engine: ctypes binding of the `mg_e1_*` entries of
include/mgrapher.h.

What the reference's model does with it (inside forward() / generate() of its transformers fork; ref: README.md:212-215,
markushgrapher/core/common/begin.py:137-151, utils/model/utils_model_loading.py:20-36): pixel_values -> `encoder.molscribe_encoder`
(MolScribe's Swin-B) -> `encoder.molscribe_projector` (MLP) -> e1 [B, 144, d_model], concatenated with the VTL encoder's states in
front of the decoder.  `E1Engine.encode(pixel_values)` returns that e1 block on the device; `Engine.attach_e1` (engine.py) makes
generate() / forward() compute it themselves.  There is no CPU fallback: without the HIP library this raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import numpy as np

from . import _lib
from .e1_shapes import E1Shape, state_dict_spec
from .engine import MgError, TorchMem


class MgE1Config(C.Structure):
    _fields_ = [("image_size", C.c_int), ("patch_size", C.c_int), ("num_channels", C.c_int), ("embed_dim", C.c_int), ("n_stages", C.c_int),
                ("depths", C.c_int * 4), ("num_heads", C.c_int * 4), ("window_size", C.c_int), ("mlp_ratio", C.c_int),
                ("layer_norm_eps", C.c_float), ("n_proj", C.c_int), ("proj_dims", C.c_int * 4), ("proj_act", C.c_int),
                ("src_image_size", C.c_int), ("pix_scale", C.c_float * 3), ("pix_shift", C.c_float * 3)]


def _config(s: E1Shape) -> MgE1Config:
    if not 1 <= len(s.depths) <= 4 or len(s.depths) != len(s.num_heads):
        raise MgError(f"e1: {len(s.depths)} stages / {len(s.num_heads)} head counts (1 .. 4 stages)")
    if len(s.proj_dims) > 3:
        raise MgError("e1: at most 4 projector layers")
    if s.proj_act not in ("gelu", "none"):
        raise MgError(f"e1: projector activation {s.proj_act!r} (gelu or none)")
    cfg = MgE1Config()
    cfg.image_size, cfg.patch_size, cfg.num_channels, cfg.embed_dim, cfg.n_stages = s.image_size, s.patch_size, s.num_channels, s.embed_dim, len(s.depths)
    for i, (d, h) in enumerate(zip(s.depths, s.num_heads)):
        cfg.depths[i], cfg.num_heads[i] = int(d), int(h)
    cfg.window_size, cfg.mlp_ratio, cfg.layer_norm_eps = s.window_size, int(s.mlp_ratio), s.layer_norm_eps
    dims = tuple(s.proj_dims) + (s.d_model,)
    cfg.n_proj = len(dims)
    for i, d in enumerate(dims):
        cfg.proj_dims[i] = int(d)
    cfg.proj_act = 1 if s.proj_act == "gelu" else 0
    cfg.src_image_size = s.src_image_size
    for i in range(3):
        cfg.pix_scale[i], cfg.pix_shift[i] = float(s.pix_scale[i]), float(s.pix_shift[i])
    return cfg


class E1Engine:
    """One set of e1 weights on the device.  `encode` only reads them, so one E1Engine serves every execution context of an engine."""

    def __init__(self, shape: E1Shape, lib=None, mem=None):
        self.lib = lib if lib is not None else _lib.load()
        self.mem = mem if mem is not None else TorchMem()
        self.shape = shape
        L = self.lib
        L.mg_last_error.restype = C.c_char_p
        L.mg_e1_weights_bytes.restype = C.c_size_t
        L.mg_e1_weights_bytes.argtypes = [C.c_void_p]
        L.mg_e1_destroy.argtypes = [C.c_void_p]
        L.mg_e1_out_tokens.argtypes = [C.c_void_p]
        L.mg_e1_bind_weights.argtypes = [C.c_void_p, C.c_void_p]
        L.mg_e1_finalize.argtypes = [C.c_void_p, C.c_void_p]
        L.mg_e1_load_tensor.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]
        L.mg_e1_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
        L.mg_e1_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        cfg = _config(shape)
        self.model = C.c_void_p()
        self._chk(L.mg_e1_create(C.byref(cfg), C.byref(self.model)))
        self.out_tokens = int(L.mg_e1_out_tokens(self.model))
        self.arena = self.mem.zeros((int(L.mg_e1_weights_bytes(self.model)),), np.uint8)
        self._chk(L.mg_e1_bind_weights(self.model, self.mem.ptr(self.arena)))
        self._ws = {}           # per host thread: a workspace is written by the call that uses it

    def _chk(self, rc):
        if rc < 0:
            raise MgError(f"libmgrapher error {rc}: {self.lib.mg_last_error().decode()}")
        return rc

    def close(self):
        if getattr(self, "model", None):
            self.lib.mg_e1_destroy(self.model)
            self.model = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd: Dict[str, object]):
        """sd: canonical keys (e1_shapes.state_dict_spec; e1_shapes.canonical_*_keys map checkpoint names onto them) -> numpy fp32 /
        uint16 bf16 bits or torch fp32 / bf16 tensors.  Every tensor of the spec is required."""
        st = self.mem.stream()
        keep = []
        want = {k for k, _, _ in state_dict_spec(self.shape)}
        for key, val in sd.items():
            if key not in want:
                raise MgError(f"e1: unexpected tensor {key!r}")
            is_bf16 = (isinstance(val, np.ndarray) and val.dtype == np.uint16) or (not isinstance(val, np.ndarray) and str(val.dtype) == "torch.bfloat16")
            h = self.mem.asarray(val, np.uint16 if is_bf16 else np.float32)
            keep.append(h)
            shp = (C.c_int64 * len(val.shape))(*[int(x) for x in val.shape])
            self._chk(self.lib.mg_e1_load_tensor(self.model, st, key.encode(), self.mem.ptr(h), 1 if is_bf16 else 0, shp, len(val.shape)))
        self._chk(self.lib.mg_e1_finalize(self.model, st))
        self.mem.sync()
        del keep
        return self

    def workspace(self, B):
        import threading
        need = C.c_size_t()
        self._chk(self.lib.mg_e1_workspace_bytes(self.model, B, C.byref(need)))
        tid = threading.get_ident()
        cur = self._ws.get(tid)
        if cur is None or cur[1] < need.value:
            self._ws[tid] = cur = (self.mem.empty((need.value,), np.uint8), need.value)
        return cur

    def encode(self, pixel_values, want_features=False):
        """pixel_values [B, 3, src, src] f32 (the VTL model's input) -> e1 [B, M, d_model] f32 on the device
        (+ SwinModel.last_hidden_state [B, M, C_last] with want_features)."""
        s = self.shape
        pv = self.mem.asarray(pixel_values, np.float32)
        B = int(pv.shape[0])
        if tuple(pv.shape) != (B, s.num_channels, s.src_image_size, s.src_image_size):
            raise ValueError(f"e1: pixel_values must be [B,{s.num_channels},{s.src_image_size},{s.src_image_size}], got {tuple(pv.shape)}")
        ws, nb = self.workspace(B)
        out = self.mem.empty((B, self.out_tokens, s.d_model), np.float32)
        feats = self.mem.empty((B, self.out_tokens, s.out_dim), np.float32) if want_features else None
        self._chk(self.lib.mg_e1_encode(self.model, self.mem.stream(), self.mem.ptr(ws), nb, self.mem.ptr(pv), B, self.mem.ptr(out),
                                        self.mem.ptr(feats) if feats is not None else None))
        self._keep = pv
        return (out, feats) if want_features else out
