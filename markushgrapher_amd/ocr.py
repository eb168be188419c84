"""ChemicalOCR stage on the MI355X engine: ctypes binding of the `mg_ocr_*` entries of
include/mgrapher.h plus the small host surface the reference uses.

Reference (markushgrapher/ocr/chemical_ocr.py:366-392):
    inputs = self.processor(text=prompt, images=[image], return_tensors="pt", size={"longest_edge": 512}).to(device)
    generated_ids = self.model.generate(**inputs, max_new_tokens=4096, do_sample=False)
    output_text = self.processor.batch_decode(generated_ids[:, prompt_len:], skip_special_tokens=True)[0]
`OcrModel.generate(input_ids=, pixel_values=, max_new_tokens=)` takes the processor's tensors and returns what
`generated_ids` holds (prompt + new tokens), so the surrounding lines stay as they are.  Tokenising, chat template and image
resizing remain the stock processor's job on the host.  There is no CPU fallback: without the HIP library this raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import numpy as np

from . import _lib
from .engine import MgError, TorchMem
from .ocr_shapes import OcrShape, PRESETS, state_dict_spec


class MgOcrConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "v_hidden", "v_inter", "v_layers", "v_heads", "image_size", "patch_size", "t_hidden", "t_inter", "t_layers", "t_heads",
        "t_kv_heads", "vocab", "scale_factor", "image_token_id", "eos_token_id", "pad_token_id", "tie_word_embeddings")] + [
        ("v_eps", C.c_float), ("rms_eps", C.c_float), ("rope_theta", C.c_float), ("n_eos_extra", C.c_int), ("eos_extra", C.c_int * 3)]


class OcrEngine:
    def __init__(self, shape: OcrShape, lib=None, mem=None):
        self.lib = lib if lib is not None else _lib.load()
        self.mem = mem if mem is not None else TorchMem()
        self.shape = shape
        L = self.lib
        L.mg_last_error.restype = C.c_char_p
        L.mg_ocr_weights_bytes.restype = C.c_size_t
        L.mg_ocr_weights_bytes.argtypes = [C.c_void_p]
        L.mg_ocr_destroy.argtypes = [C.c_void_p]
        L.mg_ocr_bind_weights.argtypes = [C.c_void_p, C.c_void_p]
        L.mg_ocr_finalize.argtypes = [C.c_void_p, C.c_void_p]
        L.mg_ocr_load_tensor.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]
        L.mg_ocr_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
        L.mg_ocr_image_features.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.mg_ocr_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p]
        L.mg_ocr_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_int]
        L.mg_ocr_stream_workspace_bytes.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.POINTER(C.c_size_t)]
        L.mg_ocr_generate_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + \
                                            [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.POINTER(C.c_long)]
        L.mg_ocr_generate_stream_ragged.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + \
                                                   [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.POINTER(C.c_long)]
        s = shape
        cfg = MgOcrConfig(s.v_hidden, s.v_inter, s.v_layers, s.v_heads, s.image_size, s.patch_size, s.t_hidden, s.t_inter, s.t_layers,
                          s.t_heads, s.t_kv_heads, s.vocab, s.scale_factor, s.image_token_id, s.eos_token_id, s.pad_token_id,
                          1 if s.tie_word_embeddings else 0, s.v_eps, s.rms_eps, s.rope_theta)
        extra = tuple(int(e) for e in getattr(s, "eos_extra", ()))
        if len(extra) > 3:
            raise MgError(f"at most 4 stop tokens are supported (eos_token_id + 3), got {1 + len(extra)}")
        cfg.n_eos_extra = len(extra)
        for i, e in enumerate(extra):
            cfg.eos_extra[i] = e
        self.model = C.c_void_p()
        self._chk(L.mg_ocr_create(C.byref(cfg), C.byref(self.model)))
        self.arena = self.mem.zeros((int(L.mg_ocr_weights_bytes(self.model)),), np.uint8)
        self._chk(L.mg_ocr_bind_weights(self.model, self.mem.ptr(self.arena)))
        self._ws = None
        self._ws_bytes = 0

    def clone(self):
        """A further execution context on this engine's weights (include/mgrapher.h mg_ocr_clone): own workspace and graphs; calls on
        different contexts may overlap when made from different host threads under different streams."""
        other = object.__new__(OcrEngine)
        other.lib, other.mem, other.shape, other.arena = self.lib, self.mem, self.shape, self.arena
        other.model = C.c_void_p()
        self.lib.mg_ocr_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        self._chk(self.lib.mg_ocr_clone(self.model, C.byref(other.model)))
        other._ws, other._ws_bytes = None, 0
        return other

    def close(self):
        if getattr(self, "model", None):
            self.lib.mg_ocr_destroy(self.model)
            self.model = None

    def __del__(self):
        try:
            if getattr(self, "model", None):
                self.lib.mg_ocr_destroy(self.model)
                self.model = None
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise MgError(f"libmgrapher error {rc}: {self.lib.mg_last_error().decode()}")
        return rc

    def load_state_dict(self, sd: Dict[str, "np.ndarray"]):
        """HF state dict of stock Idefics3ForConditionalGeneration: numpy fp32 arrays or torch tensors (fp32 / bf16)."""
        want = {k for k, _, _ in state_dict_spec(self.shape)}
        missing = sorted(want - set(sd))
        if missing:
            raise MgError(f"state dict lacks {len(missing)} tensors, e.g. {missing[:3]}")
        for k in sorted(want):
            v = sd[k]
            is_bf16 = (not isinstance(v, np.ndarray)) and str(v.dtype) == "torch.bfloat16"
            h = self.mem.asarray(v, np.uint16 if is_bf16 else np.float32)
            shape = tuple(int(x) for x in v.shape)
            arr = (C.c_int64 * len(shape))(*shape)
            self._chk(self.lib.mg_ocr_load_tensor(self.model, self.mem.stream(), k.encode(), self.mem.ptr(h), 1 if is_bf16 else 0, arr, len(shape)))
        self.mem.sync()
        self._chk(self.lib.mg_ocr_finalize(self.model, self.mem.stream()))
        self.mem.sync()
        return self

    def _workspace(self, B, n_img, L, max_new, full_logits):
        need = C.c_size_t()
        self._chk(self.lib.mg_ocr_workspace_bytes(self.model, B, n_img, L, max_new, 1 if full_logits else 0, C.byref(need)))
        if self._ws is None or self._ws_bytes < need.value:
            self._ws = None
            self._ws = self.mem.empty((int(need.value),), np.uint8)
            self._ws_bytes = need.value
        return self._ws, self._ws_bytes

    def patch_inputs(self, pixel_attention_mask):
        """pixel_attention_mask [..., I, I] (bool, as the Idefics3 processor returns it) -> (patch_pos int32 [N][P], patch_mask u8 [N][P])
        device arrays, or (None, None) when every pixel is valid.  The patch grid of get_image_features (modeling_idefics3.py:605-608)
        and the bucketed fractional coordinates of Idefics3VisionEmbeddings (:128-172), evaluated on the host with the same torch ops
        in fp32 - a few hundred integers per frame, not worth a kernel and exact by construction."""
        if pixel_attention_mask is None:
            return None, None
        import torch
        s = self.shape
        pam = torch.as_tensor(pixel_attention_mask).detach().cpu().bool()
        pam = pam.reshape(-1, s.image_size, s.image_size)
        if bool(pam.all()):
            return None, None
        ps, g = s.patch_size, s.image_size // s.patch_size
        N = pam.shape[0]
        pmask = pam.unfold(1, ps, ps).unfold(2, ps, ps).sum(dim=(-1, -2)) > 0
        boundaries = torch.arange(1 / g, 1.0, 1 / g)
        nb_h, nb_w = pmask[:, :, 0].sum(dim=1), pmask[:, 0, :].sum(dim=1)
        if bool((nb_h == 0).any()) or bool((nb_w == 0).any()):
            raise MgError("a frame without valid pixels in its first row / column (a padding image): sequences with different numbers of "
                          "frames are not supported - pass the real frames only")
        idx = torch.arange(g, dtype=torch.float32)
        fh = torch.clamp(idx[None, :] * (1.0 / nb_h)[:, None], max=(1.0 - 1e-6)).to(torch.float32)
        fw = torch.clamp(idx[None, :] * (1.0 / nb_w)[:, None], max=(1.0 - 1e-6)).to(torch.float32)
        bh, bw = torch.bucketize(fh, boundaries, right=True), torch.bucketize(fw, boundaries, right=True)
        pos = (bh[:, :, None] * g + bw[:, None, :]).reshape(N, -1)
        flat = pmask.reshape(N, -1)
        pos = torch.where(flat, pos, torch.zeros_like(pos))
        return (self.mem.asarray(pos.to(torch.int32).numpy(), np.int32), self.mem.asarray(flat.to(torch.uint8).numpy(), np.uint8))

    def _inputs(self, input_ids, pixel_values):
        ids = self.mem.asarray(input_ids, np.int64)
        B, L = int(ids.shape[0]), int(ids.shape[1])
        if pixel_values is None:
            return ids, None, B, 0, L
        pv = self.mem.asarray(pixel_values, np.float32)
        s = self.shape
        if pv.ndim != 5 or int(pv.shape[0]) != B or tuple(int(x) for x in pv.shape[2:]) != (3, s.image_size, s.image_size):
            raise MgError(f"pixel_values must be [B={B}][n_img][3][{s.image_size}][{s.image_size}], got {tuple(pv.shape)}")
        return ids, pv, B, int(pv.shape[1]), L

    def image_features(self, pixel_values, pixel_attention_mask=None):
        """[N][3][I][I] -> [N][image_seq_len][t_hidden] fp32 (get_image_features, modeling_idefics3.py:563-622)."""
        pv = self.mem.asarray(pixel_values, np.float32)
        N = int(pv.shape[0])
        pos, msk = self.patch_inputs(pixel_attention_mask)
        ws, nb = self._workspace(N, 1, 1, 0, False)
        out = self.mem.empty((N, self.shape.image_seq_len, self.shape.t_hidden), np.float32)
        self._chk(self.lib.mg_ocr_image_features(self.model, self.mem.stream(), self.mem.ptr(ws), nb, self.mem.ptr(pv),
                                                 self.mem.ptr(pos) if pos is not None else None, self.mem.ptr(msk) if msk is not None else None,
                                                 N, self.mem.ptr(out)))
        return out

    def forward_logits(self, input_ids, pixel_values=None, pixel_attention_mask=None):
        ids, pv, B, n_img, L = self._inputs(input_ids, pixel_values)
        pos, msk = self.patch_inputs(pixel_attention_mask) if pv is not None else (None, None)
        ws, nb = self._workspace(B, n_img, L, 0, True)
        out = self.mem.empty((B, L, self.shape.vocab), np.float32)
        self._chk(self.lib.mg_ocr_forward(self.model, self.mem.stream(), self.mem.ptr(ws), nb, self.mem.ptr(ids),
                                          self.mem.ptr(pv) if pv is not None else None, self.mem.ptr(pos) if pos is not None else None,
                                          self.mem.ptr(msk) if msk is not None else None, B, n_img, L, self.mem.ptr(out)))
        return out

    @staticmethod
    def left_align(input_ids, attention_mask):
        """A LEFT-PADDED batch (what the Idefics3 processor returns for prompts of different lengths: zeros of `attention_mask` in front) ->
        (ids with every row's real tokens moved to the front, prompt lengths int32 [N]).  Stock gives a real token the position
        cumsum(mask) - 1 and masks the pad keys: a row's result is that of the row alone without its padding, which is what the
        library computes from the left-aligned row and its length (mg_ocr_generate_stream_ragged)."""
        ids = np.asarray(input_ids.cpu() if hasattr(input_ids, "cpu") else input_ids, dtype=np.int64)
        am = np.asarray(attention_mask.cpu() if hasattr(attention_mask, "cpu") else attention_mask) != 0
        if ids.shape != am.shape or ids.ndim != 2:
            raise ValueError("left_align: input_ids and attention_mask are [N, L] arrays of one shape")
        lens = am.sum(axis=1).astype(np.int32)
        out = np.empty_like(ids)
        for n in range(ids.shape[0]):
            p = ids.shape[1] - int(lens[n])
            if lens[n] < 1 or am[n, :p].any() or not am[n, p:].all():
                raise ValueError(f"left_align: row {n} is not a left-padded prompt (zeros in front of the ones, at least one token)")
            out[n, :lens[n]] = ids[n, p:]
            out[n, lens[n]:] = ids[n, :p]               # (the pad tokens, behind the prompt now: never attended)
        return out, lens

    def generate_stream(self, input_ids, pixel_values=None, max_new_tokens=4096, slots=128, chunk=128, pixel_attention_mask=None, attention_mask=None):
        """Queue form (include/mgrapher.h mg_ocr_generate_stream): N pages through `slots` decode rows -> (new ids [N, max_new_tokens]
        padded after each page's stop token, lengths [N], decode steps).  Page n's ids equal generate()'s for that page.
        attention_mask [N, L] (optional): a left-padded batch of prompts of different lengths (left_align)."""
        lens = None
        if attention_mask is not None:
            input_ids, lens_np = self.left_align(input_ids, attention_mask)
            if int(lens_np.min()) < input_ids.shape[1]:
                lens = self.mem.asarray(lens_np, np.int32)
        ids, pv, N, n_img, L = self._inputs(input_ids, pixel_values)
        pos, msk = self.patch_inputs(pixel_attention_mask) if pv is not None else (None, None)
        slots, chunk = min(slots, 256), min(chunk, 256, N)
        need = C.c_size_t()
        self._chk(self.lib.mg_ocr_stream_workspace_bytes(self.model, N, n_img, L, max_new_tokens, slots, chunk, C.byref(need)))
        if getattr(self, "_sws_bytes", 0) < need.value:
            self._sws = None
            self._sws = self.mem.empty((int(need.value),), np.uint8)
            self._sws_bytes = need.value
        key = (N, max_new_tokens)
        if getattr(self, "_sout_key", None) != key:        # stable output buffers: the captured step graph holds their addresses
            self._sout = (self.mem.empty((N, max_new_tokens), np.int64), self.mem.empty((N,), np.int32))
            self._sout_key = key
        steps = C.c_long(0)
        self._chk(self.lib.mg_ocr_generate_stream_ragged(self.model, self.mem.stream(), self.mem.ptr(self._sws), self._sws_bytes, self.mem.ptr(ids),
                                                         self.mem.ptr(lens) if lens is not None else None,
                                                         self.mem.ptr(pv) if pv is not None else None, self.mem.ptr(pos) if pos is not None else None,
                                                         self.mem.ptr(msk) if msk is not None else None, N, n_img, L, max_new_tokens, slots, chunk,
                                                         self.mem.ptr(self._sout[0]), self.mem.ptr(self._sout[1]), C.byref(steps)))
        return self.mem.copy(self._sout[0]), self.mem.copy(self._sout[1]), int(steps.value)

    def generate(self, input_ids, pixel_values=None, max_new_tokens=4096, capture_steps=0, pixel_attention_mask=None):
        """-> (new_ids [B][n], step_logits or None): the tokens after the prompt, as generated_ids[:, prompt_len:] of the reference."""
        ids, pv, B, n_img, L = self._inputs(input_ids, pixel_values)
        pos, msk = self.patch_inputs(pixel_attention_mask) if pv is not None else (None, None)
        ws, nb = self._workspace(B, n_img, L, max_new_tokens, False)
        out = self.mem.empty((B, max_new_tokens), np.int64)
        cap = self.mem.empty((capture_steps, B, self.shape.vocab), np.float32) if capture_steps else None
        cols = C.c_int()
        self._chk(self.lib.mg_ocr_generate(self.model, self.mem.stream(), self.mem.ptr(ws), nb, self.mem.ptr(ids),
                                           self.mem.ptr(pv) if pv is not None else None, self.mem.ptr(pos) if pos is not None else None,
                                           self.mem.ptr(msk) if msk is not None else None, B, n_img, L, max_new_tokens, self.mem.ptr(out),
                                           C.byref(cols), self.mem.ptr(cap) if cap is not None else None, capture_steps))
        return out[:, :cols.value], cap


def shape_from_hf_config(cfg, generation_config=None) -> OcrShape:
    """OcrShape from an Idefics3 `config.json` (a dict, a path to the file, or a checkpoint directory).  The stop tokens are those
    `model.generate()` uses: `generation_config.json`'s `eos_token_id` when the checkpoint has one (an int or a list, e.g.
    <|im_end|> + <end_of_utterance>), else config.json's (generation/configuration_utils.py from_model_config)."""
    import json
    import os
    if not isinstance(cfg, dict):
        if os.path.isdir(cfg) and generation_config is None and os.path.exists(os.path.join(cfg, "generation_config.json")):
            with open(os.path.join(cfg, "generation_config.json")) as f:
                generation_config = json.load(f)
        path = os.path.join(cfg, "config.json") if os.path.isdir(cfg) else cfg
        with open(path) as f:
            cfg = json.load(f)
    v, t = cfg.get("vision_config", {}), cfg.get("text_config", {})
    rope = t.get("rope_parameters") or {}
    d = PRESETS["smoldocling"]
    eos = cfg.get("eos_token_id", t.get("eos_token_id", d.eos_token_id))
    if generation_config and generation_config.get("eos_token_id") is not None:
        eos = generation_config["eos_token_id"]
    eos_list = [int(e) for e in (eos if isinstance(eos, (list, tuple)) else [eos])]
    eos_list = list(dict.fromkeys(eos_list))
    if len(eos_list) > 4:
        raise MgError(f"{len(eos_list)} stop tokens configured; at most 4 are supported")
    eos = eos_list[0]
    if v.get("hidden_act", "gelu_pytorch_tanh") != "gelu_pytorch_tanh" or t.get("hidden_act", "silu") != "silu":
        raise MgError("unsupported activation (vision: gelu_pytorch_tanh, text: silu)")
    if t.get("model_type", "llama") != "llama" or t.get("attention_bias", False) or t.get("mlp_bias", False):
        raise MgError("unsupported text model (bias-free Llama blocks only)")
    return OcrShape(
        v_hidden=v.get("hidden_size", 1152), v_inter=v.get("intermediate_size", 3072), v_layers=v.get("num_hidden_layers", 12),
        v_heads=v.get("num_attention_heads", 16), image_size=v.get("image_size", 224), patch_size=v.get("patch_size", 32),
        v_eps=float(v.get("layer_norm_eps", 1e-6)),
        t_hidden=t.get("hidden_size", 4096), t_inter=t.get("intermediate_size", 11008), t_layers=t.get("num_hidden_layers", 32),
        t_heads=t.get("num_attention_heads", 32), t_kv_heads=t.get("num_key_value_heads", t.get("num_attention_heads", 32)),
        vocab=t.get("vocab_size", cfg.get("vocab_size", d.vocab)), rms_eps=float(t.get("rms_norm_eps", 1e-6)),
        rope_theta=float(rope.get("rope_theta", t.get("rope_theta", 10000.0))),
        scale_factor=cfg.get("scale_factor", 2), image_token_id=cfg.get("image_token_id", d.image_token_id), eos_token_id=int(eos),
        eos_extra=tuple(eos_list[1:]),
        pad_token_id=int(cfg.get("pad_token_id", t.get("pad_token_id", 0)) or 0),
        tie_word_embeddings=bool(cfg.get("tie_word_embeddings", t.get("tie_word_embeddings", False))))


class OcrModel:
    """The slice of the HF surface ChemicalOCR touches (`.generate(**inputs, max_new_tokens=, do_sample=False)`, `.eval()`, `.to()`)."""

    def __init__(self, shape: OcrShape, state_dict, device=None):
        self.engine = OcrEngine(shape, mem=TorchMem(device)).load_state_dict(state_dict)
        self.shape = shape

    @classmethod
    def from_pretrained(cls, path, device=None, **_unused):
        """Checkpoint directory of an Idefics3-class model (config.json + *.safetensors), as `AutoModelForVision2Seq.from_pretrained`
        takes it (chemical_ocr.py:76-84)."""
        import glob
        import os
        from safetensors import safe_open
        shape = shape_from_hf_config(path)
        sd = {}
        files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if not files:
            raise MgError(f"no *.safetensors under {path}")
        for fn in files:
            with safe_open(fn, framework="pt") as f:
                for k in f.keys():
                    sd[k] = f.get_tensor(k)
        if shape.tie_word_embeddings and "lm_head.weight" in sd:
            del sd["lm_head.weight"]
        return cls(shape, sd, device=device)

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def generate(self, input_ids=None, pixel_values=None, attention_mask=None, pixel_attention_mask=None, max_new_tokens=4096,
                 do_sample=False, **_unused):
        import torch
        if do_sample:
            raise MgError("only greedy search (do_sample=False) is implemented, as the reference calls it")
        if attention_mask is not None and not bool(torch.as_tensor(attention_mask).bool().all()):
            raise MgError("padded prompts are not supported (v1): batch prompts of equal length, as one page per call produces")
        new, _ = self.engine.generate(input_ids, pixel_values, max_new_tokens, pixel_attention_mask=pixel_attention_mask)
        ids = torch.as_tensor(input_ids).to(new.device)
        return torch.cat([ids, new], dim=1)
