"""Deterministic synthetic weights and inputs (no network, no checkpoints).

Everything here is a counter-based generator built from integer numpy ops only
(splitmix64 over a per-tensor FNV-1a seed), so the GPU box regenerates exactly
the same weights and inputs as the build container without shipping them.

Shapes and state-dict key names follow the UDOP layout the reference loads via
``MarkushgrapherForConditionalGeneration.from_pretrained``
(ref: markushgrapher/core/common/begin.py:130-133; key list in SURVEY.md §9.1).
Input construction follows the reference's input contract
(ref: markushgrapher/utils/common.py:34-42, core/trainers/data_collator.py:55-108,
 SURVEY.md §8(d) "Synthetic inputs").
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, asdict
from typing import Dict, List, Optional

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser (uint64 in, uint64 out)."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for c in s.encode("utf-8"):
        h ^= c
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def uniform01(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n floats in [0,1), each a multiple of 2^-24 (exact in fp32)."""
    base = np.uint64(fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF))
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) + base) & _MASK
    z = splitmix64(idx)
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)


def uniform_pm1(name: str, shape, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    return (uniform01(name, n, seed) * np.float32(2.0) - np.float32(1.0)).reshape(shape)


def randint(name: str, n: int, lo: int, hi: int, seed: int = 0) -> np.ndarray:
    """n ints in [lo, hi] inclusive."""
    base = np.uint64(fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF))
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) + base) & _MASK
    z = splitmix64(idx) >> np.uint64(11)
    return (z % np.uint64(hi - lo + 1)).astype(np.int64) + lo


def round_bf16(x: np.ndarray) -> np.ndarray:
    """Round fp32 -> bf16 (RNE) -> fp32, in pure integer numpy."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    lsb = (u >> np.uint64(16)) & np.uint64(1)
    r = (u + np.uint64(0x7FFF) + lsb) & np.uint64(0xFFFF0000)
    return r.astype(np.uint32).view(np.float32).reshape(x.shape)


# ----------------------------------------------------------------------------------------------
# model shape
# ----------------------------------------------------------------------------------------------
@dataclass
class ModelShape:
    """Subset of UdopConfig the path needs (stock:models/udop/configuration_udop.py:43-71)."""
    vocab_size: int = 33201
    d_model: int = 1024
    d_kv: int = 64
    d_ff: int = 4096
    num_layers: int = 24
    num_decoder_layers: int = 24
    num_heads: int = 16
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6
    max_2d_position_embeddings: int = 1024
    image_size: int = 512
    patch_size: int = 16
    num_channels: int = 3
    pad_token_id: int = 0
    eos_token_id: int = 1
    decoder_start_token_id: int = 0

    @property
    def num_patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2

    def to_dict(self):
        return asdict(self)


# The benchmark configuration (bench.py, tests/golden/g4_bench.npz): input seed and the recipe of the random-init weights.
# With the plain recipe (all gains 1) greedy decoding is degenerate: the tied head echoes the input token (one id repeated
# with margin ~4).  Scaling every block linear (the survey's x6) does make it varied, but only by making
# the softmax of 48 attention stacks near-argmax, i.e. the network chaotic: two fp32 implementations (stock UDOP and the
# oracle) then disagree by O(1) on the encoder output (measured: 3.6e-6 at gain 1, 6e-4 at gain 2, 2.5 at gain 4), so
# nothing could be pinned on it.  This recipe gets varied, image-dependent sequences from a numerically tame network
# instead: small token embeddings (no echo through the tied head), strong FFNs (a smooth pseudo-random function of the
# running state), moderately sharp cross-attention (image dependence); self-attention scores stay O(1).  bf16-emulated
# and fp32 oracles agree on its logits to ~0.01 (|logit| ~ 1).  Kernel cost does not depend on the values.
BENCH_SEED = 20260928
BENCH_RECIPE = dict(gain=1.0, embed_gain=0.25, ffn_gain=6.0, xq_gain=3.0)

SHAPES: Dict[str, ModelShape] = {
    # UDOP-large shape = MarkushGrapher-2's VTL encoder/decoder
    "large": ModelShape(),
    "base": ModelShape(d_model=768, d_ff=3072, num_layers=12, num_decoder_layers=12, num_heads=12),
    # fixture sizes
    # (head dim stays 64 as in every UDOP/T5 checkpoint; the HIP kernels are specialised for it)
    "tiny": ModelShape(vocab_size=500, d_model=64, d_kv=64, d_ff=128, num_layers=2, num_decoder_layers=2,
                       num_heads=2, max_2d_position_embeddings=128, image_size=64),
    "mid": ModelShape(vocab_size=2000, d_model=256, d_kv=64, d_ff=512, num_layers=4, num_decoder_layers=4,
                      num_heads=4, max_2d_position_embeddings=256, image_size=128),
}


def state_dict_spec(s: ModelShape):
    """[(hf_key, shape, kind)] for the canonical (untied) parameter set."""
    d, dk, H, dff = s.d_model, s.d_kv, s.num_heads, s.d_ff
    inner = H * dk
    nb = s.relative_attention_num_buckets
    out = [
        ("shared.weight", (s.vocab_size, d), "embed"),
        ("patch_embed.proj.weight", (d, s.num_channels, s.patch_size, s.patch_size), "conv"),
        ("patch_embed.proj.bias", (d,), "bias"),
        ("encoder.cell_2d_embedding.x_position_embeddings.weight", (s.max_2d_position_embeddings, d), "cell"),
        ("encoder.cell_2d_embedding.y_position_embeddings.weight", (s.max_2d_position_embeddings, d), "cell"),
        ("encoder.relative_bias.biases.0.relative_attention_bias.weight", (nb, H), "relbias"),
        ("encoder.relative_bias.biases.1.relative_attention_bias.weight", (nb, H), "relbias"),
        ("encoder.relative_bias.biases.2.relative_attention_bias.weight", (nb, H), "relbias"),
        ("decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", (nb, H), "relbias"),
        ("encoder.final_layer_norm.weight", (d,), "norm"),
        ("decoder.final_layer_norm.weight", (d,), "norm"),
    ]
    for i in range(s.num_layers):
        p = f"encoder.block.{i}.layer"
        out += [
            (f"{p}.0.SelfAttention.q.weight", (inner, d), "q"),
            (f"{p}.0.SelfAttention.k.weight", (inner, d), "kv"),
            (f"{p}.0.SelfAttention.v.weight", (inner, d), "kv"),
            (f"{p}.0.SelfAttention.o.weight", (d, inner), "o"),
            (f"{p}.0.layer_norm.weight", (d,), "norm"),
            (f"{p}.1.DenseReluDense.wi.weight", (dff, d), "wi"),
            (f"{p}.1.DenseReluDense.wo.weight", (d, dff), "wo"),
            (f"{p}.1.layer_norm.weight", (d,), "norm"),
        ]
    for i in range(s.num_decoder_layers):
        p = f"decoder.block.{i}.layer"
        out += [
            (f"{p}.0.SelfAttention.q.weight", (inner, d), "q"),
            (f"{p}.0.SelfAttention.k.weight", (inner, d), "kv"),
            (f"{p}.0.SelfAttention.v.weight", (inner, d), "kv"),
            (f"{p}.0.SelfAttention.o.weight", (d, inner), "o"),
            (f"{p}.0.layer_norm.weight", (d,), "norm"),
            (f"{p}.1.EncDecAttention.q.weight", (inner, d), "q"),
            (f"{p}.1.EncDecAttention.k.weight", (inner, d), "kv"),
            (f"{p}.1.EncDecAttention.v.weight", (inner, d), "kv"),
            (f"{p}.1.EncDecAttention.o.weight", (d, inner), "o"),
            (f"{p}.1.layer_norm.weight", (d,), "norm"),
            (f"{p}.2.DenseReluDense.wi.weight", (dff, d), "wi"),
            (f"{p}.2.DenseReluDense.wo.weight", (d, dff), "wo"),
            (f"{p}.2.layer_norm.weight", (d,), "norm"),
        ]
    return out


# Keys tied to canonical ones in the stock UDOP state dict
# (stock:models/udop/modeling_udop.py:1405-1413).
def tied_aliases(s: ModelShape) -> Dict[str, str]:
    return {
        "encoder.embed_tokens.weight": "shared.weight",
        "decoder.embed_tokens.weight": "shared.weight",
        "lm_head.weight": "shared.weight",
        "encoder.embed_patches.proj.weight": "patch_embed.proj.weight",
        "encoder.embed_patches.proj.bias": "patch_embed.proj.bias",
        "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight":
            "encoder.relative_bias.biases.0.relative_attention_bias.weight",
    }


def recipe_state_dict(s: ModelShape, seed: int = 20260928, gain: float = 1.0,
                      bf16_exact: bool = True, embed_gain: float = 1.0, ffn_gain: float = 1.0,
                      xq_gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Random-init weights of the given shape.  Uniform(-a, a) with a chosen per kind so activations
    stay O(1); ``gain`` scales the attention/FFN linears (default-scale random weights make
    greedy decoding collapse onto one token).  With ``bf16_exact`` every value is representable in bf16 so
    the fp32 oracle and the bf16 HIP path see identical weights.
    """
    d, dk, dff = s.d_model, s.d_kv, s.d_ff
    inner = s.num_heads * dk
    r3 = math.sqrt(3.0)
    amp = {
        "embed": embed_gain * r3,
        "conv": r3 / math.sqrt(s.num_channels * s.patch_size * s.patch_size),
        "bias": 0.1,
        "cell": 0.5 * r3,
        "relbias": 1.0 * r3,
        "q": gain * r3 / math.sqrt(d) * (dk ** -0.25),
        "kv": gain * r3 / math.sqrt(d),
        "o": gain * r3 / math.sqrt(inner),
        "wi": gain * ffn_gain * r3 / math.sqrt(d),
        "wo": gain * ffn_gain * r3 / math.sqrt(dff),
    }
    def make(item):
        key, shape, kind = item
        u = uniform_pm1(key, shape, seed)
        if kind == "norm":
            w = (np.float32(1.0) + np.float32(0.25) * u).astype(np.float32)
        else:
            a = amp[kind] * (xq_gain if key.endswith("EncDecAttention.q.weight") else 1.0)
            w = (u * np.float32(a)).astype(np.float32)
        return key, (round_bf16(w) if bf16_exact else w)

    spec = state_dict_spec(s)
    import os
    from concurrent.futures import ThreadPoolExecutor
    nthr = max(1, min(16, os.cpu_count() or 1))
    if nthr > 1 and len(spec) > 64:      # numpy releases the GIL inside the integer kernels
        with ThreadPoolExecutor(max_workers=nthr) as ex:
            return dict(ex.map(make, spec))
    return dict(make(it) for it in spec)


# ----------------------------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------------------------
def synth_pages_u8(B: int, size: int = 1024, seed: int = 20260928) -> np.ndarray:
    """White size×size RGB canvases with random black segments and text-box-like bars
    (chemical-page-like sparsity).  uint8 [B,size,size,3]."""
    img = np.full((B, size, size, 3), 255, dtype=np.uint8)
    for b in range(B):
        nseg = int(randint(f"page{b}.nseg", 1, 20, 60, seed)[0])
        p = randint(f"page{b}.seg", nseg * 4, 0, size - 1, seed).reshape(nseg, 4)
        for x0, y0, x1, y1 in p:
            n = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
            xs = np.linspace(x0, x1, n).round().astype(np.int64)
            ys = np.linspace(y0, y1, n).round().astype(np.int64)
            for dx in (0, 1):
                img[b, np.clip(ys + dx, 0, size - 1), np.clip(xs, 0, size - 1)] = 0
        nbox = int(randint(f"page{b}.nbox", 1, 8, 40, seed)[0])
        q = randint(f"page{b}.box", nbox * 2, 0, size - 64, seed).reshape(nbox, 2)
        for x0, y0 in q:
            img[b, y0:y0 + 12:2, x0:x0 + 48] = 40
    return img


def pages_to_pixel_values(pages_u8: np.ndarray, image_size: int) -> np.ndarray:
    """1024² u8 HWC crops -> model input [B,3,image_size,image_size] f32 in [-1,1]
    (reference: LANCZOS resize ref:mdu_dataset.py:118 then x/255, mean=std=0.5,
    SURVEY.md §8(a) a12).  The synthetic bench uses an exact box filter for the integer down-scale
    (content is irrelevant to cost); the preprocessing stage itself is "next" row f-3."""
    B, Hh, Ww, C = pages_u8.shape
    f = Hh // image_size
    x = pages_u8.astype(np.float32)
    if f > 1:
        x = x.reshape(B, image_size, f, image_size, f, C).mean(axis=(2, 4))
    x = (x / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)).astype(np.float32)


def synth_batch(s: ModelShape, B: int, L_min: int = 32, L_max: int = 256, seed: int = 20260928,
                page_px: Optional[int] = None, fixed_L: Optional[int] = None, return_pages: bool = False):
    """Batch of model inputs in the reference's contract:
    input_ids [B,L] i64, bbox [B,L,4] f32 in [0,1], attention_mask [B,L] i64, pixel_values [B,3,I,I] f32.
    Text: 12 question tokens with box 0, one sep with box 1, OCR sub-words with word boxes, one sep with
    box 1; rows padded to the batch max with id 0 / box 0 / mask 0
    (ref: core/trainers/data_collator.py:55-61,63-103)."""
    if fixed_L is not None:
        lens = np.full(B, fixed_L, dtype=np.int64)
    else:
        lens = randint("batch.len", B, L_min, L_max, seed)
    L = int(lens.max())
    ids = np.zeros((B, L), dtype=np.int64)
    bbox = np.zeros((B, L, 4), dtype=np.float32)
    mask = np.zeros((B, L), dtype=np.int64)
    vmax = min(31999, s.vocab_size - 1)
    for b in range(B):
        n = int(lens[b])
        ids[b, :n] = randint(f"b{b}.ids", n, 3, vmax, seed)
        mask[b, :n] = 1
        nq = min(12, max(n - 2, 0))
        u = uniform01(f"b{b}.box", n * 4, seed).reshape(n, 4)
        x0 = u[:, 0] * np.float32(0.9)
        y0 = u[:, 1] * np.float32(0.9)
        w = np.float32(0.01) + u[:, 2] * np.float32(0.07)
        h = np.float32(0.01) + u[:, 3] * np.float32(0.02)
        bb = np.stack([x0, y0, np.clip(x0 + w, 0, 1), np.clip(y0 + h, 0, 1)], axis=-1).astype(np.float32)
        bb[:nq] = 0.0
        if n > nq:
            bb[nq] = 1.0
            ids[b, nq] = s.eos_token_id
        if n > nq + 1:
            bb[n - 1] = 1.0
            ids[b, n - 1] = s.eos_token_id
        bbox[b, :n] = bb
    px = page_px if page_px is not None else 2 * s.image_size
    pages = synth_pages_u8(B, px, seed)
    pixel_values = pages_to_pixel_values(pages, s.image_size)
    out = {"input_ids": ids, "bbox": bbox, "attention_mask": mask, "pixel_values": pixel_values}
    if return_pages:
        out["pages_u8"] = pages
    return out
