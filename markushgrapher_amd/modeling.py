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
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from .engine import Engine, TorchMem
from .synth import ModelShape, state_dict_spec, tied_aliases


class MarkushgrapherConfig:
    """UdopConfig-compatible field names (stock transformers models/udop/configuration_udop.py:43-71) plus the
    attributes the reference sets (`architecture_variant`, `output_attentions`, ref: begin.py:119-121)."""

    model_type = "markushgrapher"

    def __init__(self, **kw):
        base = ModelShape()
        for k, v in base.to_dict().items():
            setattr(self, k, kw.pop(k, v))
        self.num_decoder_layers = kw.pop("num_decoder_layers", None) or self.num_layers
        self.architecture_variant = kw.pop("architecture_variant", "me-lf-stack-1")
        self.output_attentions = kw.pop("output_attentions", False)
        self.max_length = kw.pop("max_length", 512)
        self.tie_word_embeddings = True
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
        d.update(architecture_variant=self.architecture_variant, model_type=self.model_type)
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


class _Placeholder(nn.Module):
    """`encoder.molscribe_encoder` / `encoder.molscribe_projector` (OCSR e1 branch): the fork's
    Swin-B branch is not part of this path; the attribute names exist so the reference's helpers do not fail."""


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
        self.lm_head.add("weight", self._tree._modules["shared"].weight.data)     # tied (stock:1412)
        self.encoder.add_module("molscribe_encoder", _Placeholder())
        self.encoder.add_module("molscribe_projector", _Placeholder())
        del self._modules["_tree"]

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
        for k, v in state_dict.items():
            ck = k if k == "lm_head.weight" else aliases.get(k, k)
            if ck in own and ck != "lm_head.weight":
                if tuple(own[ck].shape) != tuple(v.shape):
                    raise ValueError(f"shape mismatch for {k}: {tuple(v.shape)} vs {tuple(own[ck].shape)}")
                own[ck].copy_(torch.as_tensor(v).to(own[ck].dtype))
                seen.add(ck)
            elif k == "lm_head.weight":
                t = torch.as_tensor(v).to(self._weight_dtype).to(self.lm_head.weight.device)
                if torch.equal(t, self.shared.weight.data) or "shared.weight" not in state_dict:
                    self.lm_head.weight.data = self.shared.weight.data        # tied
                else:
                    self.lm_head.weight.data = t
                seen.add(k)
            elif not (k.startswith("decoder.embed_patches") or k.startswith("decoder.relative_bias")
                      or k.startswith("encoder.molscribe_")):
                unexpected.append(k)
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

    def init_molscribe_weights(self):                        # ref: begin.py:137-138 — e1 branch is outside this path
        return None

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
        save_file({k: v.contiguous() for k, v in self._canonical_items()}, os.path.join(path, "model.safetensors"))

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
            eng = Engine(self._shape, mem=TorchMem(dev), max_decode_len=max(512, int(self.config.max_length)))
            sd = {k: v.data for k, v in self._canonical_items()}
            lm = self.lm_head.weight.data
            if lm.data_ptr() != self.shared.weight.data_ptr() and not torch.equal(lm.to(dev), self.shared.weight.data):
                sd["lm_head.weight"] = lm       # untied head in the checkpoint (ref: utils_model_loading.py:41)
            eng.load_state_dict(sd)
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
                decoder_input_ids=None, decoder_attention_mask=None, **kw):
        """stock:1448-1574.  Returns logits [B,T,V] fp32 (+ CE loss when labels are given)."""
        eng = self._eng()
        if decoder_input_ids is None:
            if labels is None:
                raise ValueError("forward() needs labels or decoder_input_ids")
            decoder_input_ids = self._shift_right(labels)
        logits, enc, mask = eng.forward_logits(input_ids, bbox, attention_mask, pixel_values, decoder_input_ids,
                                               decoder_attention_mask)
        loss = None
        if labels is not None:
            loss = nn.functional.cross_entropy(logits.view(-1, logits.size(-1)), labels.to(logits.device).view(-1),
                                               ignore_index=-100)
        return Seq2SeqLMOutput(loss=loss, logits=logits, encoder_last_hidden_state=enc, encoder_attention_mask=mask)

    @torch.no_grad()
    def generate(self, input_ids=None, bbox=None, pixel_values=None, attention_mask=None, labels=None, num_beams=1,
                 max_length=None, min_length=0, length_penalty=1.0, early_stopping=False, do_sample=False, **kw):
        """ref call: utils_evaluation.py:269-285 (`labels` arrives as a stray kwarg and is ignored).  Without an
        attention_mask the fork's transformers 4.34 base infers one from pad tokens (all ones at the reference's batch
        size 1), which is what is reproduced here; pass a mask explicitly for padded batches."""
        if do_sample:
            raise NotImplementedError("sampling is not part of the reference's decode path")
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
                                 early_stopping=early_stopping)
        return ids
