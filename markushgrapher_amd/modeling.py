"""HuggingFace-style surface of the MarkushGrapher-2 VTL encoder + CXSMILES decoder on the MI355X HIP engine.

Mirrors what the reference imports from its transformers fork (`transformers.models.markushgrapher`,
ref: markushgrapher/core/common/begin.py:7-13) for THIS path, with the same names, argument meaning and outputs:

    config = MarkushgrapherConfig.from_pretrained(path); config.image_size = 512        (ref: begin.py:114-121)
    model  = MarkushgrapherForConditionalGeneration.from_pretrained(path, config=config).to(device)   (begin.py:128-133)
    ids    = model.generate(**encoding, num_beams=5, max_length=512)     (ref: utils/ocsr/utils_evaluation.py:269-285)
    logits = model(**sample).logits                                      (ref: core/trainers/curriculumTrainer.py:654-656)

The torch tensors held here are only the HF-layout copy of the weights (for state_dict()/parameters(), which the
reference's save/compare helpers walk: ref: utils/model/utils_model_loading.py:6-46) and the memory carrier of
inputs/outputs; every computation of forward()/generate() runs in the HIP library behind the C ABI.
"""
from __future__ import annotations

import json
import os
import warnings
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from .engine import Engine, MgError, TorchMem
from .synth import ModelShape, state_dict_spec, tied_aliases


class MarkushgrapherConfig:
    """UdopConfig-compatible field names (stock transformers models/udop/configuration_udop.py:43-71) plus the
    attributes the reference sets (`architecture_variant`, `output_attentions`, ref: begin.py:119-121)."""

    model_type = "markushgrapher"

    def __init__(self, **kw):
        base = ModelShape()
        ndl = kw.get("num_decoder_layers", None)                 # UdopConfig: defaults to num_layers when absent / None
        for k, v in base.to_dict().items():
            setattr(self, k, kw.pop(k, v))
        self.num_decoder_layers = ndl if ndl is not None else self.num_layers
        # "none" = the plain VTL encoder + decoder; "me-lf-stack-1" = MarkushGrapher-2's two-encoder late fusion, whose OCSR
        # branch (e1) this package does not compute (ref default: core/common/arguments.py:258; set at begin.py:120)
        self.architecture_variant = kw.pop("architecture_variant", "none")
        self.allow_missing_e1 = bool(kw.pop("allow_missing_e1", False))
        # geometry of the OCSR vision branch: overrides of e1_shapes.PRESETS["swin_b_384"] (MolScribe's Swin-B), e.g. {"pix_scale": [...]};
        # d_model / src_image_size follow this config, the projector's layer sizes follow the checkpoint's tensors
        self.e1 = dict(kw.pop("e1", None) or {})
        self.output_attentions = kw.pop("output_attentions", False)
        self.max_length = kw.pop("max_length", 512)
        self.tie_word_embeddings = bool(kw.pop("tie_word_embeddings", True))      # UDOP / T5 default
        self.is_encoder_decoder = True
        self.extra = kw

    @classmethod
    def from_pretrained(cls, path, **kw):
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        d.update(kw)
        if isinstance(d.get("image_size"), (list, tuple)):
            d["image_size"] = d["image_size"][0]
        return cls(**d)

    def to_shape(self) -> ModelShape:
        return ModelShape(**{k: getattr(self, k) for k in ModelShape().to_dict()})

    def to_dict(self):
        d = self.to_shape().to_dict()
        d.update(architecture_variant=self.architecture_variant, model_type=self.model_type,
                 tie_word_embeddings=self.tie_word_embeddings, max_length=self.max_length)
        if self.e1:
            d["e1"] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in self.e1.items()}
        return d


@dataclass
class Seq2SeqLMOutput:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    encoder_last_hidden_state: Optional[torch.Tensor] = None
    encoder_attention_mask: Optional[torch.Tensor] = None

    def __getitem__(self, i):
        return (self.loss, self.logits)[i] if self.loss is not None else (self.logits,)[i]


class _ParamTree(nn.Module):
    """nn.Module tree built from dotted state-dict keys, so `.encoder.block[0]...`, `.decoder`, `.lm_head` exist with
    the HF names and carry state_dict()/parameters()."""

    def add(self, dotted: str, tensor: torch.Tensor):
        head, _, rest = dotted.partition(".")
        if not rest:
            self.register_parameter(head, nn.Parameter(tensor, requires_grad=False))
            return
        if head not in self._modules:
            self.add_module(head, _ParamTree())
        self._modules[head].add(rest, tensor)


class _E1Branch(_ParamTree):
    """`encoder.molscribe_encoder` / `encoder.molscribe_projector` (OCSR e1 branch).  The modules are tensor
    containers with the checkpoint's own names and dtypes, so `.state_dict()` / `.parameters()` / save_weights_separately
    (ref: utils/model/utils_model_loading.py:6-46) and `model.safe_load(model.encoder.molscribe_projector, states)`
    (ref: begin.py:151) round-trip them unchanged; the computation - Swin encoder + projector on the HIP engine (e1.E1Engine,
    csrc/swin.hip) - is set up from these tensors when the model is first used (e1_shapes.canonical_*_keys maps timm /
    transformers namings onto the library's).  With an incomplete branch the tokens have to be supplied as `e1=`."""

    def load_state_dict(self, state_dict, strict=False):
        self._modules.clear()
        self._parameters.clear()
        for k, v in state_dict.items():
            self.add(k, torch.as_tensor(v).detach().clone())
        return [], []


class MarkushgrapherForConditionalGeneration(nn.Module):
    main_input_name = "input_ids"

    def __init__(self, config: MarkushgrapherConfig, weight_dtype=torch.bfloat16):
        super().__init__()
        self.config = config
        self._shape = config.to_shape()
        self._weight_dtype = weight_dtype
        self._engine: Optional[Engine] = None
        self._engine_device = None
        self._tree = _ParamTree()
        for key, shp, _ in state_dict_spec(self._shape):
            self._tree.add(key, torch.zeros(shp, dtype=weight_dtype))
        self.add_module("encoder", self._tree._modules["encoder"])
        self.add_module("decoder", self._tree._modules["decoder"])
        self.add_module("shared", self._tree._modules["shared"])
        self.add_module("patch_embed", self._tree._modules["patch_embed"])
        self.lm_head = _ParamTree()
        shared_w = self._tree._modules["shared"].weight.data
        self.lm_head.add("weight", shared_w if config.tie_word_embeddings else shared_w.clone())     # tied: stock:1412
        self.encoder.add_module("molscribe_encoder", _E1Branch())
        self.encoder.add_module("molscribe_projector", _E1Branch())
        del self._modules["_tree"]
        self.ignored_keys = []                 # keys of the last load_state_dict that nothing on this path consumes
        self._warned_e1 = False

    # ---- weights ------------------------------------------------------------------------------------------------
    def _canonical_items(self):
        aliases = tied_aliases(self._shape)
        own = dict(super().state_dict())
        for key, _, _ in state_dict_spec(self._shape):
            yield key, own[key]

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = False):
        aliases = tied_aliases(self._shape)
        own = {k: v for k, v in super().state_dict().items()}
        missing, unexpected = [], []
        seen = set()
        self.ignored_keys = []
        e1_parts = {"encoder.molscribe_encoder.": {}, "encoder.molscribe_projector.": {}}
        for k, v in state_dict.items():
            pref = next((p for p in e1_parts if k.startswith(p)), None)
            if pref is not None:                      # kept on the HF side (round trip), not consumed by the HIP engine
                e1_parts[pref][k[len(pref):]] = v
                continue
            ck = k if k == "lm_head.weight" else aliases.get(k, k)
            if ck in own and ck != "lm_head.weight":
                if tuple(own[ck].shape) != tuple(v.shape):
                    raise ValueError(f"shape mismatch for {k}: {tuple(v.shape)} vs {tuple(own[ck].shape)}")
                own[ck].copy_(torch.as_tensor(v).to(own[ck].dtype))
                seen.add(ck)
            elif k == "lm_head.weight":
                if self.config.tie_word_embeddings:
                    # HF re-ties the head to shared.weight after loading (tie_weights): the checkpoint's tensor is dropped
                    self.lm_head.weight.data = self.shared.weight.data
                    self.ignored_keys.append(k)
                else:
                    self.lm_head.weight.data = torch.as_tensor(v).to(self._weight_dtype).to(self.lm_head.weight.device)
                seen.add(k)
            elif k.startswith("decoder.embed_patches") or k.startswith("decoder.relative_bias"):
                self.ignored_keys.append(k)           # present in UDOP state dicts, never used by the decoder (stock:1212-1213)
            else:
                unexpected.append(k)
        if e1_parts["encoder.molscribe_encoder."]:
            self.encoder.molscribe_encoder.load_state_dict(e1_parts["encoder.molscribe_encoder."])
        if e1_parts["encoder.molscribe_projector."]:
            self.encoder.molscribe_projector.load_state_dict(e1_parts["encoder.molscribe_projector."])
        for key, _, _ in state_dict_spec(self._shape):
            if key not in seen:
                missing.append(key)
        if strict and (missing or unexpected):
            raise RuntimeError(f"missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        self._engine = None     # re-pushed to the HIP arena on next use
        return missing, unexpected

    def safe_load(self, module: nn.Module, state_dict):      # ref: begin.py:151,166
        module.load_state_dict(state_dict, strict=False)
        self._engine = None

    MOLSCRIBE_CKPT = os.path.join("external", "MolScribe", "ckpts", "swin_base_char_aux_1m680k.pth")     # ref: setup.sh:79-82

    def init_molscribe_weights(self, path: Optional[str] = None):
        """ref: begin.py:137-138 - loads the pretrained MolScribe Swin-B (`swin_base_char_aux_1m680k.pth`, ref: setup.sh:79-82) into
        `encoder.molscribe_encoder`.  The checkpoint is looked up at `path`, $MG_MOLSCRIBE_CKPT, or the reference's install location
        relative to the working directory; its `encoder` entry (MolScribe saves {'encoder': ..., 'decoder': ...}, keys possibly behind
        `module.`) is kept under timm's own names.  Without the file the branch keeps what the model checkpoint held (the published
        MarkushGrapher-2 weights carry their frozen copy) and the caller is told."""
        cand = [p for p in (path, os.environ.get("MG_MOLSCRIBE_CKPT"), self.MOLSCRIBE_CKPT) if p]
        hit = next((p for p in cand if os.path.exists(p)), None)
        if hit is None:
            have = len(self.encoder.molscribe_encoder.state_dict())
            warnings.warn(f"markushgrapher_amd: init_molscribe_weights(): no MolScribe checkpoint at {cand}; encoder.molscribe_encoder keeps the "
                          f"{have} tensors of the model checkpoint", stacklevel=2)
            return
        ck = torch.load(hit, map_location="cpu", weights_only=True)
        enc = ck.get("encoder", ck) if isinstance(ck, dict) else ck
        enc = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in enc.items()}
        self.encoder.molscribe_encoder.load_state_dict(enc)
        self._engine = None

    def _e1_setup(self):
        """(E1Shape, canonical state dict) of the OCSR branch when both modules hold a complete set of tensors, else None."""
        from . import e1_shapes
        enc_sd = self.encoder.molscribe_encoder.state_dict()
        prj_sd = self.encoder.molscribe_projector.state_dict()
        if not enc_sd or not prj_sd:
            return None
        import dataclasses
        canon = dict(e1_shapes.canonical_encoder_keys(enc_sd))
        canon.update(e1_shapes.canonical_projector_keys(prj_sd))
        over = {k: (tuple(v) if isinstance(v, list) else v) for k, v in self.config.e1.items()}
        base = dataclasses.replace(e1_shapes.PRESETS["swin_b_384"], d_model=self.config.d_model, src_image_size=self.config.image_size, **over)
        shape = e1_shapes.shape_from_state(base, canon)
        if shape.d_model != self.config.d_model:
            raise ValueError(f"encoder.molscribe_projector ends in {shape.d_model} features, d_model is {self.config.d_model}")
        want = {k: tuple(shp) for k, shp, _ in e1_shapes.state_dict_spec(shape)}
        missing = [k for k in want if k not in canon]
        if missing:
            raise ValueError(f"encoder.molscribe_*: {len(missing)} tensors of the OCSR branch are missing, e.g. {missing[:3]}")
        bad = [k for k in want if tuple(canon[k].shape) != want[k]]
        if bad:
            raise ValueError(f"encoder.molscribe_*: shape mismatch for {bad[:3]} (expected {[want[k] for k in bad[:3]]})")
        return shape, {k: canon[k] for k in want}

    def computes_e1(self) -> bool:
        """True when `encoder.molscribe_encoder` and `encoder.molscribe_projector` hold a complete branch: forward() / generate() then
        evaluate it themselves."""
        return len(self.encoder.molscribe_encoder.state_dict()) > 0 and len(self.encoder.molscribe_projector.state_dict()) > 0

    def requires_e1(self) -> bool:
        """True when the configured architecture fuses the OCSR branch: variant 'me-lf-stack-*' (ref: config/predict.yaml:12,
        utils_model_loading.py:20) or a checkpoint that carries `encoder.molscribe_*` tensors."""
        return str(self.config.architecture_variant).startswith("me-lf-stack") or \
            len(self.encoder.molscribe_encoder.state_dict()) > 0 or len(self.encoder.molscribe_projector.state_dict()) > 0

    def _check_e1(self, e1):
        if e1 is not None or not self.requires_e1():
            return
        if self.computes_e1():
            try:
                self._e1_setup()
            except (KeyError, ValueError) as err:
                raise RuntimeError(f"the encoder.molscribe_* tensors do not form a usable OCSR branch (e1): {err}. Load the complete "
                                   "branch or pass the projected embeddings as e1=[B, M, d_model].") from err
            return
        msg = (f"architecture_variant={self.config.architecture_variant!r}: this model fuses the OCSR vision branch (e1), but "
               "encoder.molscribe_encoder / encoder.molscribe_projector hold no weights to compute it from (load a checkpoint that "
               "carries them, or init_molscribe_weights() + safe_load(model.encoder.molscribe_projector, ...)). Alternatively pass "
               "the projected embeddings as e1=[B, M, d_model]; running without them differs from the reference.")
        if self.config.allow_missing_e1 or os.environ.get("MG_ALLOW_MISSING_E1") == "1":
            if not self._warned_e1:
                warnings.warn(msg + " (allow_missing_e1: continuing with the VTL branch only)", stacklevel=3)
                self._warned_e1 = True
            return
        raise RuntimeError(msg + " Set config.allow_missing_e1 = True (or MG_ALLOW_MISSING_E1=1) to run the VTL branch alone.")

    @classmethod
    def from_pretrained(cls, path, config: Optional[MarkushgrapherConfig] = None, **kw):
        config = config or MarkushgrapherConfig.from_pretrained(path)
        model = cls(config)
        sd = {}
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        model.load_state_dict(sd, strict=False)
        return model.eval()

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.config.to_dict(), f, indent=1)
        from safetensors.torch import save_file
        out = {k: v.contiguous() for k, v in self._canonical_items()}
        if not self.config.tie_word_embeddings:
            out["lm_head.weight"] = self.lm_head.weight.data.contiguous()
        for name in ("molscribe_encoder", "molscribe_projector"):          # the e1 branch's tensors travel with the checkpoint
            for k, v in getattr(self.encoder, name).state_dict().items():
                out[f"encoder.{name}.{k}"] = v.contiguous()
        save_file(out, os.path.join(path, "model.safetensors"))

    # ---- engine -------------------------------------------------------------------------------------------------
    @property
    def device(self):
        return self.shared.weight.device

    def get_encoder(self):
        return self.encoder

    def _eng(self) -> Engine:
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("MarkushgrapherForConditionalGeneration runs on an MI355X only: call .to('cuda') first "
                               "(there is no CPU fallback)")
        if self._engine is None or self._engine_device != dev:
            tied = bool(self.config.tie_word_embeddings)
            eng = Engine(self._shape, mem=TorchMem(dev), max_decode_len=max(512, int(self.config.max_length)),
                         tie_word_embeddings=tied)
            sd = {k: v.data for k, v in self._canonical_items()}
            if not tied:                        # untied config: the checkpoint's head is used, without the d_model^-0.5 scale
                sd["lm_head.weight"] = self.lm_head.weight.data
            eng.load_state_dict(sd)
            if self.computes_e1():              # the OCSR branch on the HIP engine, attached: calls without e1= evaluate it themselves
                from .e1 import E1Engine
                try:
                    shape1, sd1 = self._e1_setup()
                    e1e = E1Engine(shape1, mem=TorchMem(dev)).load_state_dict({k: v.data for k, v in sd1.items()})
                    eng.attach_e1(e1e)
                except (KeyError, ValueError, MgError) as exc:
                    # an incomplete / unsupported branch: the engine stays unattached, so that a caller-supplied e1= still works
                    # (a call WITHOUT e1= then fails in _check_e1 with the reason)
                    self._e1_attach_error = str(exc)
                    warnings.warn(f"OCSR branch (encoder.molscribe_*) not attached to the HIP engine: {exc}; pass e1= explicitly", stacklevel=2)
                else:
                    # What the fork does AROUND the Swin encoder is not in the reference tree (its transformers fork is absent): say once which
                    # choices this build made, so that outputs differing from the reference's are not silent.
                    warnings.warn(
                        "OCSR vision branch attached from the checkpoint's encoder.molscribe_* tensors.  The Swin encoder is pinned on stock "
                        "transformers SwinModel; the steps around it are INFERRED from the reference's README (fork source absent) and unpinned: "
                        f"input = bilinear resize of pixel_values {shape1.src_image_size} -> {shape1.image_size} px with per-channel affine "
                        f"scale {tuple(shape1.pix_scale)} / shift {tuple(shape1.pix_shift)} (MolScribe itself trains on ImageNet mean/std: set "
                        "config.e1 = dict(markushgrapher_amd.e1_shapes.IMAGENET_RENORM) if your checkpoint expects that), projector activation "
                        f"'{shape1.proj_act}', decoder keys = [e1 | VTL states].  Pass e1= to supply the tokens yourself.", stacklevel=2)
            self._engine, self._engine_device = eng, dev
        return self._engine

    def _shift_right(self, labels):                          # stock:791-811
        out = labels.new_zeros(labels.shape)
        out[..., 1:] = labels[..., :-1].clone()
        out[..., 0] = self.config.decoder_start_token_id
        out.masked_fill_(out == -100, self.config.pad_token_id)
        return out

    @torch.no_grad()
    def forward(self, input_ids=None, bbox=None, attention_mask=None, pixel_values=None, labels=None,
                decoder_input_ids=None, decoder_attention_mask=None, e1=None, **kw):
        """stock:1448-1574.  Returns logits [B,T,V] fp32 (+ CE loss when labels are given).  e1: optional [B, M, d_model]
        embeddings of the OCSR vision branch (see _E1Branch)."""
        self._check_e1(e1)
        eng = self._eng()
        if decoder_input_ids is None:
            if labels is None:
                raise ValueError("forward() needs labels or decoder_input_ids")
            decoder_input_ids = self._shift_right(labels)
        logits, enc, mask = eng.forward_logits(input_ids, bbox, attention_mask, pixel_values, decoder_input_ids,
                                               decoder_attention_mask, e1=e1)
        loss = None
        if labels is not None:
            loss = nn.functional.cross_entropy(logits.view(-1, logits.size(-1)), labels.to(logits.device).view(-1),
                                               ignore_index=-100)
        return Seq2SeqLMOutput(loss=loss, logits=logits, encoder_last_hidden_state=enc, encoder_attention_mask=mask)

    @torch.no_grad()
    def generate(self, input_ids=None, bbox=None, pixel_values=None, attention_mask=None, labels=None, num_beams=1,
                 max_length=None, min_length=0, length_penalty=1.0, early_stopping=False, do_sample=False, e1=None, **kw):
        """ref call: utils_evaluation.py:269-285 (`labels` arrives as a stray kwarg and is ignored).  Without an
        attention_mask the fork's transformers 4.34 base infers one from pad tokens (all ones at the reference's batch
        size 1), which is what is reproduced here; pass a mask explicitly for padded batches."""
        if do_sample:
            raise NotImplementedError("sampling is not part of the reference's decode path")
        self._check_e1(e1)
        eng = self._eng()
        max_length = int(max_length or self.config.max_length)
        if attention_mask is None:
            pad, eos = self.config.pad_token_id, self.config.eos_token_id
            if pad is not None and pad != eos and bool((input_ids == pad).any()):
                attention_mask = (input_ids != pad).long()
            else:
                attention_mask = torch.ones_like(input_ids)
        ids, _, _ = eng.generate(input_ids, bbox, attention_mask, pixel_values, num_beams=int(num_beams),
                                 max_length=max_length, min_length=int(min_length), length_penalty=float(length_penalty),
                                 early_stopping=early_stopping, e1=e1)
        return ids

    @torch.no_grad()
    def in_flight(self, n: int = 4):
        """`inflight.InFlight` over this model's engine: n execution contexts (the engine + n - 1 clones on the same weights), a host
        thread and stream each, for keeping several batches going at once: `with model.in_flight(4) as fl: fl.map(job, batches)`
        where `job(ctx, batch)` calls `ctx.preprocess` / `ctx.generate` (engine level: arrays in, ids out).  DESIGN.md section 8f."""
        from .inflight import InFlight
        self._check_e1(None)
        return InFlight(self._eng(), n)

    def generate_queue(self, encodings, max_length=None, min_length=0, slots=32, chunk=32, contexts=1, num_beams=1, length_penalty=1.0,
                       early_stopping=False):
        """The reference's evaluation loop (ref: utils/ocsr/utils_evaluation.py:140-285) as ONE call: `encodings` = the per-sample
        dicts it builds (input_ids [1, L_n] or [L_n], bbox, pixel_values; attention_mask / labels ignored as there), greedy,
        max_length as there.  Returns a list of 1-D id tensors - predictions[n] == self.generate(**encodings[n], num_beams=1,
        max_length=max_length)[0] - decoded by the continuous decoder (mg_generate_stream: `slots` rows work through the queue, a row
        that emits EOS hands its slot to the next image) with per-image padding semantics (every image computed as if alone).
        contexts > 1: the queue is cut into that many contiguous parts, each decoded by its own execution context (inflight.InFlight,
        at most 4) at the same time - same ids, 1.4 x the images/s at 4.
        num_beams > 1 (the reference's shipped setting, config/predict.yaml beam_search: True -> 5): the beam queue (mg_generate_stream_beam,
        `slots` image slots of num_beams rows, slots * num_beams <= 256); predictions[n] == self.generate(**encodings[n],
        num_beams=num_beams, ...)[0]."""
        from .assembly import collate_for_generate
        self._check_e1(None)
        eng = self._eng()
        max_length = int(max_length or self.config.max_length)
        feats = []
        for e in encodings:
            ids = torch.as_tensor(e["input_ids"]).reshape(-1).cpu()
            feats.append({"input_ids": ids, "bbox": torch.as_tensor(e["bbox"]).reshape(-1, 4).cpu().to(torch.float32)})
        batch = collate_for_generate(feats)
        pix = torch.cat([torch.as_tensor(e["pixel_values"]).reshape(1, *torch.as_tensor(e["pixel_values"]).shape[-3:]) for e in encodings]).to(self.device)
        n = len(feats)
        num_beams = int(num_beams)
        if num_beams > 1:
            slots = max(1, min(int(slots), 256 // num_beams))

        def run(ctx, sl):
            m = sl.stop - sl.start
            if num_beams > 1:
                o, l, _, _ = ctx.generate_stream_beam(batch["input_ids"][sl], batch["bbox"][sl], batch["attention_mask"][sl], pix[sl],
                                                      num_beams=num_beams, max_length=max_length, min_length=int(min_length),
                                                      length_penalty=float(length_penalty), early_stopping=bool(early_stopping),
                                                      chunk=min(chunk, m), slots=min(slots, m), pool_chunks=3)
            else:
                o, l, _ = ctx.generate_stream(batch["input_ids"][sl], batch["bbox"][sl], batch["attention_mask"][sl], pix[sl],
                                              max_length=max_length, min_length=int(min_length), chunk=min(chunk, m), slots=min(slots, m),
                                              pool_chunks=3)
            return o, l
        contexts = max(1, min(int(contexts), 4, n // max(1, min(slots, n))))
        prev = eng.set_padding_semantics(True)
        try:
            if contexts > 1:
                from .inflight import InFlight
                if getattr(self, "_inflight", None) is None or len(self._inflight) != contexts:
                    if getattr(self, "_inflight", None) is not None:
                        self._inflight.close()
                    self._inflight = InFlight(eng, contexts, include_source=False)      # clones made under per-image padding semantics
                    for c in self._inflight.contexts:
                        c.set_stream_encoder(0)                                         # the contexts overlap one another instead
                per = -(-n // contexts)
                torch.cuda.current_stream(self.device).synchronize()      # the inputs are complete before the contexts' streams read them

                def part(ctx, i):
                    return run(ctx, slice(i * per, min(n, (i + 1) * per)))
                outs = self._inflight.map(part, range(-(-n // per)))
                rows = []
                for o, l in outs:
                    l = l.cpu().tolist()
                    rows += [o[i, :l[i]] for i in range(len(l))]
                return rows
            ids, lens = run(eng, slice(0, n))
        finally:
            eng.set_padding_semantics(prev)
        lens = lens.cpu().tolist()
        return [ids[i, :lens[i]] for i in range(len(lens))]

