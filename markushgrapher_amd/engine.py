"""Thin ctypes binding of libmgrapher_hip.so (include/mgrapher.h).  PyTorch-ROCm is only the memory carrier:
device buffers are torch tensors, every computation is a HIP kernel behind the C ABI.

`Engine` is written against a tiny memory-provider interface so the parity tests can also drive the SAME C ABI
compiled for the CPU SIMT emulator with numpy buffers (tests/backends.py) — the product default, `TorchMem`, is
the only provider in this package and requires a GPU; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

from . import _lib
from .synth import ModelShape


class MgConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "vocab_size", "d_model", "d_kv", "d_ff", "num_layers", "num_decoder_layers", "num_heads",
        "relative_attention_num_buckets", "relative_attention_max_distance", "max_2d_position_embeddings",
        "image_size", "patch_size", "num_channels", "pad_token_id", "eos_token_id", "decoder_start_token_id")] + [
        ("layer_norm_epsilon", C.c_float), ("max_decode_len", C.c_int), ("tie_word_embeddings", C.c_int)]


class MgError(RuntimeError):
    pass


class TorchMem:
    """Device memory carried by torch tensors on one GPU."""

    def __init__(self, device=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("markushgrapher_amd needs an MI355X GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.torch = torch
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")

    _NP2T = None

    def _dt(self, dtype):
        t = self.torch
        return {np.dtype(np.float32): t.float32, np.dtype(np.float64): t.float64, np.dtype(np.int64): t.int64,
                np.dtype(np.int32): t.int32, np.dtype(np.uint8): t.uint8, np.dtype(np.uint16): t.int16}[np.dtype(dtype)]

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=self._dt(dtype), device=self.device)

    def zeros(self, shape, dtype):
        return self.torch.zeros(shape, dtype=self._dt(dtype), device=self.device)

    def asarray(self, x, dtype):
        """numpy array or torch tensor -> contiguous device tensor of `dtype` (no copy when already right)."""
        t = self.torch
        if isinstance(x, np.ndarray):
            if x.dtype == np.uint16:
                x = x.view(np.int16)
            x = t.from_numpy(np.ascontiguousarray(x))
        want = self._dt(dtype)
        if x.dtype == t.bfloat16 and np.dtype(dtype) == np.uint16:
            x = x.view(t.int16)
        # host sources are copied synchronously: a non-blocking copy from pageable memory may read the (temporary) source after this
        # function has returned and released it (the runtime's queue-thread mode, AMD_DIRECT_DISPATCH=0, does exactly that)
        return x.to(device=self.device, dtype=want, non_blocking=x.is_cuda).contiguous()

    def copy(self, h):
        return h.clone()

    def ptr(self, h):
        return C.c_void_p(h.data_ptr())

    def stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def sync(self):
        # the calling thread's current stream only: a device-wide synchronisation is refused by the runtime ("operation not permitted
        # when stream is capturing") whenever ANOTHER execution context is capturing its decode step at that moment
        self.torch.cuda.current_stream(self.device).synchronize()

    def numpy(self, h):
        self.sync()
        return h.detach().cpu().numpy()


class Engine:
    MAX_LIVE_ROWS = 256     # B * num_beams per generate() call (include/mgrapher.h "Limits")

    def __init__(self, shape: ModelShape, lib=None, mem=None, max_decode_len: int = 512, tie_word_embeddings: bool = True):
        self.lib = lib if lib is not None else _lib.load()
        self.mem = mem if mem is not None else TorchMem()
        self.shape = shape
        self._declare()
        cfg = MgConfig(shape.vocab_size, shape.d_model, shape.d_kv, shape.d_ff, shape.num_layers,
                       shape.num_decoder_layers, shape.num_heads, shape.relative_attention_num_buckets,
                       shape.relative_attention_max_distance, shape.max_2d_position_embeddings, shape.image_size,
                       shape.patch_size, shape.num_channels, shape.pad_token_id, shape.eos_token_id,
                       shape.decoder_start_token_id, shape.layer_norm_epsilon, max_decode_len,
                       1 if tie_word_embeddings else 0)
        self.model = C.c_void_p()
        self._chk(self.lib.mg_create(C.byref(cfg), C.byref(self.model)))
        self.max_decode_len = (max_decode_len + 63) // 64 * 64
        self.arena = self.mem.zeros((int(self.lib.mg_weights_bytes(self.model)),), np.uint8)
        self._chk(self.lib.mg_bind_weights(self.model, self.mem.ptr(self.arena)))
        self._ws = None
        self._ws_bytes = 0
        self._gen_out = {}
        self.ignored_keys = []
        self._e1_engine = None

    def _declare(self):
        L = self.lib
        L.mg_last_error.restype = C.c_char_p
        L.mg_weights_bytes.restype = C.c_size_t
        L.mg_weights_bytes.argtypes = [C.c_void_p]
        L.mg_destroy.argtypes = [C.c_void_p]
        L.mg_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.mg_bind_weights.argtypes = [C.c_void_p, C.c_void_p]
        L.mg_finalize.argtypes = [C.c_void_p, C.c_void_p]
        L.mg_load_tensor.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]
        L.mg_set_decode_graph.argtypes = [C.c_void_p, C.c_int]
        L.mg_attach_e1.argtypes = [C.c_void_p, C.c_void_p]
        L.mg_set_shared_gpu.argtypes = [C.c_void_p, C.c_int]
        L.mg_set_cross_absorb.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mg_decode_graph_active.argtypes = [C.c_void_p]
        L.mg_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
        L.mg_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.mg_decoder_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_void_p]
        L.mg_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                  C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        L.mg_stream_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
        L.mg_generate_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + \
                                        [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.POINTER(C.c_long)]
        L.mg_stream_beam_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
        L.mg_generate_stream_beam.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + \
                                             [C.c_int] * 8 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_long)]
        L.mg_stream_encoder_mode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.mg_debug_bucket_table.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.mg_debug_decode_capture.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]

    def _chk(self, rc):
        if rc < 0:
            raise MgError(f"libmgrapher error {rc}: {self.lib.mg_last_error().decode()}")
        return rc

    def set_decode_graph(self, mode):
        """0: eager launches; 1: captured decode-step HIP graph (default); 2: the graph's device-counter kernels, eager."""
        return int(self.lib.mg_set_decode_graph(self.model, int(mode)))

    def decode_graph_active(self):
        return bool(self.lib.mg_decode_graph_active(self.model))

    def set_shared_gpu(self, shared: bool) -> bool:
        """Other execution contexts run beside this one (include/mgrapher.h mg_set_shared_gpu): the cross-attention K/V stream keeps one
        workgroup per CU resident so that the other contexts' small launches find wave slots.  InFlight sets it on its contexts.  Returns the
        previous setting."""
        return bool(self.lib.mg_set_shared_gpu(self.model, 1 if shared else 0))

    def set_cross_absorb(self, absorb, key_splits: int = 0):
        """Greedy cross-attention form (include/mgrapher.h mg_set_cross_absorb): True / 1 = weight-absorbed (the layers stream the attended
        encoder states), False / 0 = per-layer K / V streams, "auto" / 2 (default) = by the call's decode rows (absorbed from 96 on).  The
        workspace is re-sized at the next call.  Returns the previous setting (False / True / "auto")."""
        mode = 2 if absorb == "auto" else int(absorb)
        prev = self._chk(self.lib.mg_set_cross_absorb(self.model, mode, int(key_splits)))
        self._ws, self._ws_bytes = None, 0
        return "auto" if prev == 2 else bool(prev)

    @property
    def cross_absorb(self):
        """False / True / "auto": the cross-attention form greedy calls of this context run (mg_set_cross_absorb)."""
        prev = int(self.lib.mg_set_cross_absorb(self.model, -1, 0))
        return "auto" if prev == 2 else bool(prev)

    ABSORB_AUTO_ROWS = 96      # "auto": calls with at least this many decode rows take the absorbed form (engine.hip MG_ABSORB_AUTO_ROWS)

    def clone(self):
        """A further execution context on this engine's weights (include/mgrapher.h mg_clone): same arena, own workspace, decode
        graph and output buffers.  Calls on different contexts may overlap in time when made from different host threads, each under
        its own stream (`with torch.cuda.stream(s): ctx.generate(...)`); ids per batch are those of a call made alone."""
        other = object.__new__(Engine)
        other.lib, other.mem, other.shape = self.lib, self.mem, self.shape
        other.max_decode_len = self.max_decode_len
        other.arena = self.arena            # shared, kept alive by every context
        other.model = C.c_void_p()
        self._chk(self.lib.mg_clone(self.model, C.byref(other.model)))
        other._ws, other._ws_bytes, other._gen_out, other.ignored_keys = None, 0, {}, list(self.ignored_keys)
        other._e1_engine = self._e1_engine      # mg_clone carries the attachment; the branch's weights are shared and kept alive
        return other

    def attach_e1(self, e1_engine):
        """Attach the OCSR vision branch (e1.E1Engine; None detaches): generate() / encode() / the queue forms called without `e1=` then
        evaluate it from pixel_values inside the call and decode over [e1 | e2] (include/mgrapher.h mg_attach_e1) - the reference's
        `architecture_variant: me-lf-stack-1`.  Clones made afterwards inherit it."""
        self.mem.sync()
        self._chk(self.lib.mg_attach_e1(self.model, e1_engine.model if e1_engine is not None else None))
        self._e1_engine = e1_engine
        self._ws, self._ws_bytes, self._gen_out = None, 0, {}
        self._sws, self._sws_bytes = None, 0
        return self

    def release_workspaces(self):
        """Drop the cached workspaces and output buffers of this context (they are re-allocated by the next call that needs them).  A
        workspace is sized by the largest call the context has seen (14 GB for 64 rows x 512 positions at the large shape): a process
        that moves on to another stage with its own contexts gives the memory back first.  The captured decode graph replays on the
        workspace it was captured on, so the library forgets it as well (mg_set_decode_graph re-arms the capture)."""
        self.mem.sync()
        self._ws, self._ws_bytes, self._gen_out = None, 0, {}
        self._sws, self._sws_bytes = None, 0
        if self.model:
            prev = self.lib.mg_set_decode_graph(self.model, 0)
            self.lib.mg_set_decode_graph(self.model, prev)

    def close(self):
        if self.model:
            self.lib.mg_destroy(self.model)
            self.model = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, object], finalize: bool = True):
        """sd: HF state-dict names -> numpy (fp32 / uint16 bf16 bits) or torch tensors (fp32 / bf16)."""
        st = self.mem.stream()
        keep = []
        for key, val in sd.items():
            is_bf16 = (isinstance(val, np.ndarray) and val.dtype == np.uint16) or \
                      (not isinstance(val, np.ndarray) and str(val.dtype) == "torch.bfloat16")
            h = self.mem.asarray(val, np.uint16 if is_bf16 else np.float32)
            keep.append(h)
            shp = (C.c_int64 * len(val.shape))(*[int(s) for s in val.shape])
            rc = self._chk(self.lib.mg_load_tensor(self.model, st, key.encode(), self.mem.ptr(h), 1 if is_bf16 else 0,
                                                   shp, len(val.shape)))
            if rc == 1:
                self.ignored_keys.append(key)
        if finalize:
            self._chk(self.lib.mg_finalize(self.model, st))
        self.mem.sync()
        del keep

    def bucket_table(self, which: int, n: int) -> np.ndarray:
        out = (C.c_int * n)()
        k = self.lib.mg_debug_bucket_table(self.model, which, out, n)
        return np.array(out[:k], dtype=np.int32)

    def workspace(self, B, L, num_beams, max_length, T, M_e1=0):
        need = C.c_size_t()
        self._chk(self.lib.mg_workspace_bytes(self.model, B, L, num_beams, max_length, T, M_e1, C.byref(need)))
        if need.value > self._ws_bytes:
            self._ws = None
            self._ws = self.mem.empty((need.value,), np.uint8)
            self._ws_bytes = need.value
        return self._ws, self._ws_bytes

    def preprocess(self, pages_u8):
        """u8 pages [B][Hs][Ws][3] (RGB) -> pixel_values [B][3][I][I] f32 on the device: Pillow-LANCZOS resize to the
        model's input size + 1/255 + (x-0.5)/0.5, bit-exact with the reference's host preprocessing
        (ref: core/datasets/mdu_dataset.py:118, core/common/begin.py:105-109)."""
        L = self.lib
        L.mg_preprocess_scratch_bytes.restype = C.c_size_t
        L.mg_preprocess_scratch_bytes.argtypes = [C.c_int] * 4
        L.mg_preprocess_pages.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        pg = self.mem.asarray(pages_u8, np.uint8)
        B, Hs, Ws, ch = [int(v) for v in pg.shape]
        if ch != self.shape.num_channels:
            raise ValueError("pages must be [B, H, W, 3] uint8")
        I = self.shape.image_size
        nb = int(L.mg_preprocess_scratch_bytes(B, Hs, Ws, I))
        if getattr(self, "_prep_bytes", 0) < nb:
            self._prep = self.mem.empty((nb,), np.uint8)
            self._prep_bytes = nb
        out = self.mem.empty((B, ch, I, I), np.float32)
        self._chk(L.mg_preprocess_pages(self.mem.stream(), self.mem.ptr(pg), B, Hs, Ws, I, self.mem.ptr(out),
                                        self.mem.ptr(self._prep), self._prep_bytes))
        return out

    def _inputs(self, input_ids, bbox, attention_mask, pixel_values):
        ids = self.mem.asarray(input_ids, np.int64)
        bb = self.mem.asarray(bbox, np.float32)
        pv = self.mem.asarray(pixel_values, np.float32)
        am = None if attention_mask is None else self.mem.asarray(attention_mask, np.uint8)
        B, L = int(ids.shape[0]), int(ids.shape[1])
        s = self.shape
        if tuple(bb.shape) != (B, L, 4):
            raise ValueError(f"bbox must be [B,L,4], got {tuple(bb.shape)}")
        if tuple(pv.shape) != (B, s.num_channels, s.image_size, s.image_size):
            raise ValueError(f"pixel_values must be [B,{s.num_channels},{s.image_size},{s.image_size}], got {tuple(pv.shape)}")
        return ids, bb, am, pv, B, L

    def _e1(self, e1, B):
        """e1 [B, M, d_model] fp32: precomputed embeddings of the OCSR vision branch (include/mgrapher.h mg_encode), or None."""
        if e1 is None:
            return None, 0
        t = self.mem.asarray(e1, np.float32)
        if len(t.shape) != 3 or int(t.shape[0]) != B or int(t.shape[2]) != self.shape.d_model or int(t.shape[1]) < 1:
            raise ValueError(f"e1 must be [B, M, {self.shape.d_model}], got {tuple(t.shape)}")
        return t, int(t.shape[1])

    def encode(self, input_ids, bbox, attention_mask, pixel_values, max_length=0, num_beams=1, T=0, want_out=True, e1=None):
        ids, bb, am, pv, B, L = self._inputs(input_ids, bbox, attention_mask, pixel_values)
        e1t, M = self._e1(e1, B)
        ws, nb = self.workspace(B, L, num_beams, max_length, T, M)
        S = L + self.shape.num_patches
        out = self.mem.empty((B, S, self.shape.d_model), np.float32) if want_out else None
        msk = self.mem.empty((B, S), np.uint8) if want_out else None
        self._chk(self.lib.mg_encode(self.model, self.mem.stream(), self.mem.ptr(ws), nb, self.mem.ptr(ids), self.mem.ptr(bb),
                                     self.mem.ptr(am) if am is not None else None, self.mem.ptr(pv),
                                     self.mem.ptr(e1t) if e1t is not None else None, M, B, L,
                                     self.mem.ptr(out) if want_out else None, self.mem.ptr(msk) if want_out else None))
        self._keep = (ids, bb, am, pv, e1t)
        return out, msk

    def forward_logits(self, input_ids, bbox, attention_mask, pixel_values, decoder_input_ids, decoder_attention_mask=None, e1=None):
        dec = self.mem.asarray(decoder_input_ids, np.int64)
        B, T = int(dec.shape[0]), int(dec.shape[1])
        dm = None if decoder_attention_mask is None else self.mem.asarray(decoder_attention_mask, np.uint8)
        enc_out, enc_mask = self.encode(input_ids, bbox, attention_mask, pixel_values, T=T, e1=e1)
        ws, nb = self.workspace(B, int(enc_out.shape[1]) - self.shape.num_patches, 1, 0, T, 0 if e1 is None else int(e1.shape[1]))
        logits = self.mem.empty((B, T, self.shape.vocab_size), np.float32)
        self._chk(self.lib.mg_decoder_forward(self.model, self.mem.stream(), self.mem.ptr(ws), nb, self.mem.ptr(dec),
                                              self.mem.ptr(dm) if dm is not None else None, B, T, self.mem.ptr(logits)))
        return logits, enc_out, enc_mask

    def debug_decode_capture(self, capture_steps=0, rows=0, forced_ids=None):
        """Parity-test instrumentation (include/mgrapher.h mg_debug_decode_capture): returns the device buffer
        [capture_steps][rows][vocab] that the next generate() calls fill with their per-step logits; forced_ids
        [B][max_length] teacher-forces the decode path.  Call with no arguments to clear."""
        cap = self.mem.zeros((capture_steps, rows, self.shape.vocab_size), np.float32) if capture_steps > 0 else None
        frc = None if forced_ids is None else self.mem.asarray(forced_ids, np.int64)
        self._dbg_keep = (cap, frc)
        self._chk(self.lib.mg_debug_decode_capture(self.model, self.mem.ptr(cap) if cap is not None else None, capture_steps,
                                                   self.mem.ptr(frc) if frc is not None else None))
        return cap

    def set_padding_semantics(self, per_image: bool):
        """True: every image of a padded batch is computed as if it were alone and unpadded (the reference's batch size is 1);
        False (default): stock HF batched semantics (include/mgrapher.h mg_set_padding_semantics)."""
        self.lib.mg_set_padding_semantics.argtypes = [C.c_void_p, C.c_int]
        return bool(self.lib.mg_set_padding_semantics(self.model, 1 if per_image else 0))

    def set_stream_encoder(self, mode=1, cu_mask=None):
        """Where generate_stream's encoder runs: 0 the caller's stream (serial), 1 its own low-priority stream, 2 its own stream on
        the compute units of `cu_mask` (iterable of CU bit indices)."""
        if mode == 2:
            bits = sorted(set(int(b) for b in cu_mask))
            nw = bits[-1] // 32 + 1
            arr = (C.c_uint32 * nw)()
            for b in bits:
                arr[b // 32] |= 1 << (b % 32)
            self._chk(self.lib.mg_stream_encoder_mode(self.model, 2, arr, nw))
        else:
            self._chk(self.lib.mg_stream_encoder_mode(self.model, int(mode), None, 0))

    def generate_stream(self, input_ids, bbox, attention_mask, pixel_values, max_length=512, min_length=0, chunk=32, slots=32,
                        pool_chunks=3):
        """Continuous greedy decoding of N images (include/mgrapher.h mg_generate_stream): -> (ids [N, max_length] padded with the pad
        id, lengths [N] = valid columns, decode steps run).  Row n equals generate()'s row for image n."""
        ids, bb, am, pv, N, L = self._inputs(input_ids, bbox, attention_mask, pixel_values)
        need = C.c_size_t()
        self._chk(self.lib.mg_stream_workspace_bytes(self.model, chunk, L, slots, pool_chunks, C.byref(need)))
        if getattr(self, "_sws_bytes", 0) < need.value:
            self._sws = None
            self._sws = self.mem.empty((need.value,), np.uint8)
            self._sws_bytes = need.value
        okey = ("stream", N, max_length)
        out = self._gen_out.get(okey)
        if out is None:
            out = self._gen_out[okey] = (self.mem.empty((N, max_length), np.int64), self.mem.empty((N,), np.int32))
        steps = C.c_long(0)
        self._chk(self.lib.mg_generate_stream(self.model, self.mem.stream(), self.mem.ptr(self._sws), self._sws_bytes, self.mem.ptr(ids),
                                              self.mem.ptr(bb), self.mem.ptr(am) if am is not None else None, self.mem.ptr(pv), N, L, chunk, slots,
                                              pool_chunks, max_length, min_length, self.mem.ptr(out[0]), self.mem.ptr(out[1]), C.byref(steps)))
        return self.mem.copy(out[0]), self.mem.copy(out[1]), int(steps.value)

    def generate_stream_beam(self, input_ids, bbox, attention_mask, pixel_values, num_beams=5, max_length=512, min_length=0,
                             length_penalty=1.0, early_stopping=False, chunk=32, slots=32, pool_chunks=3):
        """Continuous BEAM-SEARCH decoding of N images (include/mgrapher.h mg_generate_stream_beam): `slots` image slots of num_beams
        rows -> (ids [N, max_length] = best hypothesis per image, lengths [N], scores [N], decode steps run).  Row n equals
        generate(num_beams=...)'s result for image n."""
        ids, bb, am, pv, N, L = self._inputs(input_ids, bbox, attention_mask, pixel_values)
        if slots * num_beams > self.MAX_LIVE_ROWS:
            raise MgError(f"generate_stream_beam: slots * num_beams = {slots * num_beams} rows exceed the supported {self.MAX_LIVE_ROWS}")
        need = C.c_size_t()
        self._chk(self.lib.mg_stream_beam_workspace_bytes(self.model, chunk, L, slots, pool_chunks, num_beams, max_length, C.byref(need)))
        if getattr(self, "_sws_bytes", 0) < need.value:
            self._sws = None
            self._sws = self.mem.empty((need.value,), np.uint8)
            self._sws_bytes = need.value
        okey = ("stream-beam", N, max_length)
        out = self._gen_out.get(okey)
        if out is None:
            out = self._gen_out[okey] = (self.mem.empty((N, max_length), np.int64), self.mem.empty((N,), np.int32),
                                         self.mem.empty((N,), np.float32))
        steps = C.c_long(0)
        self._chk(self.lib.mg_generate_stream_beam(self.model, self.mem.stream(), self.mem.ptr(self._sws), self._sws_bytes, self.mem.ptr(ids),
                                                   self.mem.ptr(bb), self.mem.ptr(am) if am is not None else None, self.mem.ptr(pv), N, L, chunk,
                                                   slots, pool_chunks, num_beams, max_length, min_length, C.c_float(length_penalty),
                                                   1 if early_stopping else 0, self.mem.ptr(out[0]), self.mem.ptr(out[1]),
                                                   self.mem.ptr(out[2]), C.byref(steps)))
        return self.mem.copy(out[0]), self.mem.copy(out[1]), self.mem.copy(out[2]), int(steps.value)

    def generate(self, input_ids, bbox, attention_mask, pixel_values, num_beams=1, max_length=512, min_length=0,
                 length_penalty=1.0, early_stopping=False, return_top2=False, e1=None):
        ids, bb, am, pv, B, L = self._inputs(input_ids, bbox, attention_mask, pixel_values)
        e1t, M = self._e1(e1, B)
        if B * num_beams > self.MAX_LIVE_ROWS:
            raise MgError(f"generate: B * num_beams = {B * num_beams} live sequences exceeds the supported "
                          f"{self.MAX_LIVE_ROWS}; split the batch")
        ws, nb = self.workspace(B, L, num_beams, max_length, 0, M)
        # the id buffer is persistent per (B, max_length): the captured decode-step graph holds its address, so a stable
        # buffer lets later calls replay the graph instead of re-capturing; callers get a copy
        okey = (B, max_length)
        out = self._gen_out.get(okey)
        if out is None:
            self._gen_out.clear()
            out = self._gen_out[okey] = self.mem.empty((B, max_length), np.int64)
        scores = self.mem.zeros((B,), np.float32)
        top2 = self.mem.zeros((max_length, B * num_beams, 2), np.float32) if (return_top2 and num_beams == 1) else None
        cols = C.c_int(0)
        self._chk(self.lib.mg_generate(self.model, self.mem.stream(), self.mem.ptr(ws), nb, self.mem.ptr(ids), self.mem.ptr(bb),
                                       self.mem.ptr(am) if am is not None else None, self.mem.ptr(pv),
                                       self.mem.ptr(e1t) if e1t is not None else None, M, B, L, num_beams,
                                       max_length, min_length, C.c_float(length_penalty), 1 if early_stopping is True else 0,
                                       self.mem.ptr(out), C.byref(cols), self.mem.ptr(scores),
                                       self.mem.ptr(top2) if top2 is not None else None))
        return self.mem.copy(out[:, :cols.value]), scores, top2
