"""Shapes, state-dict layout, key maps and deterministic recipe weights of the OCSR vision branch "e1" (SURVEY.md §8 rows a7 / f-2).

What the reference says about the branch (its source lives in the un-vendored transformers fork):
  * "The input image is processed by an OCSR vision encoder (Swin-B ViT, from MolScribe) followed by an MLP projector"
    (ref: README.md:212-215); the two encoders' outputs are concatenated in front of the decoder ("late fusion", variant
    `me-lf-stack-1`, ref: config/predict.yaml:12).
  * module names `model.encoder.molscribe_encoder` / `model.encoder.molscribe_projector`
    (ref: markushgrapher/utils/model/utils_model_loading.py:20-36, core/common/begin.py:151),
    `model.init_molscribe_weights()` loads `swin_base_char_aux_1m680k.pth` (ref: begin.py:137-138, setup.sh:79-82).
  * MolScribe's encoder is timm 0.4.12 `swin_base_patch4_window12_384` (ref: setup.py:39): patch 4, embed 128, depths 2/2/18/2,
    heads 4/8/16/32, window 12, input 384 x 384 -> 144 tokens of 1024 features after the final LayerNorm.

The Swin-B arithmetic is pinned on stock `transformers.SwinModel` of this geometry (86.88 M parameters, [B, 144, 1024]); canonical
key names below are stock transformers 5.15 `SwinModel` names.  INFERRED (the fork's `modeling_markushgrapher.py` is absent) and
therefore configuration here, not constants: how the branch's 384-px input is derived from the model's 512-px `pixel_values`
(`src_image_size`, bilinear resize, optional per-channel re-normalisation `pix_scale` / `pix_shift`), the projector's layer sizes
and activation (`proj_dims`, `proj_act`).
"""
from __future__ import annotations

import re
from dataclasses import dataclass, asdict
from typing import Dict, List, Tuple

import numpy as np

from .synth import uniform_pm1, round_bf16


@dataclass(frozen=True)
class E1Shape:
    # Swin encoder (SwinConfig names)
    image_size: int = 384
    patch_size: int = 4
    num_channels: int = 3
    embed_dim: int = 128
    depths: Tuple[int, ...] = (2, 2, 18, 2)
    num_heads: Tuple[int, ...] = (4, 8, 16, 32)
    window_size: int = 12
    mlp_ratio: int = 4
    layer_norm_eps: float = 1e-5
    # projector: Linear(C_last -> proj_dims[0]) -> act -> ... -> Linear(-> d_model); () = one Linear
    proj_dims: Tuple[int, ...] = (1024,)
    proj_act: str = "gelu"            # "gelu" (erf form) or "none"
    d_model: int = 1024
    # the branch's input is derived from the VTL model's pixel_values [B, 3, src, src]: bilinear resize to image_size, then
    # x * pix_scale[c] + pix_shift[c]  (identity by default; mean = std = 0.5 -> ImageNet statistics would be
    # scale = 0.5 / std_c, shift = (0.5 - mean_c) / std_c)
    src_image_size: int = 512
    pix_scale: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    pix_shift: Tuple[float, float, float] = (0.0, 0.0, 0.0)

    @property
    def n_stages(self) -> int:
        return len(self.depths)

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    def stage_dim(self, i: int) -> int:
        return self.embed_dim << i

    def stage_res(self, i: int) -> int:
        return self.grid >> i

    def stage_window(self, i: int) -> int:
        """Effective window of stage i (stock modeling_swin.py:576-582: clamped to the resolution, then no shift)."""
        return min(self.window_size, self.stage_res(i))

    @property
    def out_tokens(self) -> int:
        return self.stage_res(self.n_stages - 1) ** 2

    @property
    def out_dim(self) -> int:
        return self.stage_dim(self.n_stages - 1)

    def as_dict(self):
        return asdict(self)


IMAGENET_RENORM = dict(pix_scale=(0.5 / 0.229, 0.5 / 0.224, 0.5 / 0.225),
                       pix_shift=((0.5 - 0.485) / 0.229, (0.5 - 0.456) / 0.224, (0.5 - 0.406) / 0.225))

PRESETS: Dict[str, E1Shape] = {
    # MolScribe's Swin-B (timm swin_base_patch4_window12_384) + a 2-layer MLP projector into UDOP-large's d_model
    "swin_b_384": E1Shape(),
    # parity fixture: every code path of the big one (shifted windows with the region mask, three stages, two merges, a last stage
    # whose window is the whole map) at a size the CPU oracle and the SIMT emulator finish in seconds; pairs with synth.SHAPES["tiny"]
    "tiny": E1Shape(image_size=64, embed_dim=64, depths=(2, 2, 2), num_heads=(2, 4, 8), window_size=4, proj_dims=(128,), d_model=64,
                    src_image_size=64),
    # window 12 at a small size (GPU fixture; the emulator takes minutes on it): 96 px -> 24 x 24 -> 12 x 12
    "w12": E1Shape(image_size=96, embed_dim=64, depths=(2, 2), num_heads=(2, 4), window_size=12, proj_dims=(128,), d_model=64,
                   src_image_size=128),
}


def state_dict_spec(s: E1Shape) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(canonical key, shape, kind): `swin.*` = stock SwinModel (add_pooling_layer=False) names, `proj.{j}.*` = the projector's j-th
    Linear.  kind in {conv, linear, bias, norm_w, norm_b, relpos}."""
    out = []
    e = "swin.embeddings."
    out.append((e + "patch_embeddings.projection.weight", (s.embed_dim, s.num_channels, s.patch_size, s.patch_size), "conv"))
    out.append((e + "patch_embeddings.projection.bias", (s.embed_dim,), "bias"))
    out.append((e + "norm.weight", (s.embed_dim,), "norm_w"))
    out.append((e + "norm.bias", (s.embed_dim,), "norm_b"))
    for i in range(s.n_stages):
        C = s.stage_dim(i)
        H = s.num_heads[i]
        w = s.window_size             # the table keeps the configured window even where the effective window is clamped (stock:413)
        for j in range(s.depths[i]):
            p = f"swin.encoder.layers.{i}.blocks.{j}."
            for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
                out.append((p + f"attention.{n}.weight", (C, C), "linear"))
                out.append((p + f"attention.{n}.bias", (C,), "bias"))
            out.append((p + "attention.relative_position_bias.relative_position_bias_table", ((2 * w - 1) ** 2, H), "relpos"))
            out.append((p + "layernorm_before.weight", (C,), "norm_w"))
            out.append((p + "layernorm_before.bias", (C,), "norm_b"))
            out.append((p + "layernorm_after.weight", (C,), "norm_w"))
            out.append((p + "layernorm_after.bias", (C,), "norm_b"))
            out.append((p + "mlp.fc1.weight", (s.mlp_ratio * C, C), "linear"))
            out.append((p + "mlp.fc1.bias", (s.mlp_ratio * C,), "bias"))
            out.append((p + "mlp.fc2.weight", (C, s.mlp_ratio * C), "linear"))
            out.append((p + "mlp.fc2.bias", (C,), "bias"))
        if i + 1 < s.n_stages:
            p = f"swin.encoder.layers.{i}.downsample."
            out.append((p + "reduction.weight", (2 * C, 4 * C), "linear"))
            out.append((p + "norm.weight", (4 * C,), "norm_w"))
            out.append((p + "norm.bias", (4 * C,), "norm_b"))
    out.append(("swin.layernorm.weight", (s.out_dim,), "norm_w"))
    out.append(("swin.layernorm.bias", (s.out_dim,), "norm_b"))
    dims = (s.out_dim,) + tuple(s.proj_dims) + (s.d_model,)
    for j in range(len(dims) - 1):
        out.append((f"proj.{j}.weight", (dims[j + 1], dims[j]), "linear"))
        out.append((f"proj.{j}.bias", (dims[j + 1],), "bias"))
    return out


def recipe_state_dict(s: E1Shape, seed: int = 20260930, gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Counter-based weights (bf16-exact fp32; regenerated identically on the GPU box, nothing stored): linear / conv
    U(-1,1) gain sqrt(3 / fan_in), biases 0.1 U, norm weights 1 + 0.1 U, norm biases 0.1 U, relative-position tables U (so that
    the bias and the shift mask matter in the softmax)."""
    sd = {}
    for name, shape, kind in state_dict_spec(s):
        u = uniform_pm1("e1/" + name, shape, seed)
        if kind in ("linear", "conv"):
            fan_in = int(np.prod(shape[1:]))
            w = u * np.float32(gain * np.sqrt(3.0 / fan_in))
        elif kind == "norm_w":
            w = np.float32(1.0) + np.float32(0.1) * u
        elif kind == "relpos":
            w = u
        else:
            w = np.float32(0.1) * u
        sd[name] = round_bf16(w.astype(np.float32))
    return sd


def synth_pixels(s: E1Shape, B: int, seed: int = 20260930) -> np.ndarray:
    """[B, 3, src, src] f32 in [-1, 1]: smooth low-frequency content plus noise (a resize of pure noise would test little)."""
    n = s.src_image_size
    yy, xx = np.meshgrid(np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32), indexing="ij")
    out = np.empty((B, 3, n, n), np.float32)
    for b in range(B):
        for c in range(3):
            ph = uniform_pm1(f"e1pix.{b}.{c}", (4,), seed)
            wave = np.sin(xx * np.float32(0.05 + 0.04 * ph[0]) + ph[1] * 3.0) * np.cos(yy * np.float32(0.04 + 0.03 * ph[2]) + ph[3] * 3.0)
            out[b, c] = np.float32(0.6) * wave + np.float32(0.4) * uniform_pm1(f"e1pix.n.{b}.{c}", (n, n), seed)
    return np.clip(out, -1.0, 1.0).astype(np.float32)


# ----------------------------------------------------------------------------------------------------------------------------
# checkpoint key names -> canonical names
# ----------------------------------------------------------------------------------------------------------------------------
_TIMM_BLOCK = [
    (r"norm1\.(weight|bias)$", r"layernorm_before.\1"),
    (r"norm2\.(weight|bias)$", r"layernorm_after.\1"),
    (r"attn\.proj\.(weight|bias)$", r"attention.o_proj.\1"),
    (r"attn\.relative_position_bias_table$", r"attention.relative_position_bias.relative_position_bias_table"),
    (r"mlp\.fc([12])\.(weight|bias)$", r"mlp.fc\1.\2"),
]
_HF4_BLOCK = [
    (r"attention\.self\.query\.(weight|bias)$", r"attention.q_proj.\1"),
    (r"attention\.self\.key\.(weight|bias)$", r"attention.k_proj.\1"),
    (r"attention\.self\.value\.(weight|bias)$", r"attention.v_proj.\1"),
    (r"attention\.self\.relative_position_bias_table$", r"attention.relative_position_bias.relative_position_bias_table"),
    (r"attention\.output\.dense\.(weight|bias)$", r"attention.o_proj.\1"),
    (r"intermediate\.dense\.(weight|bias)$", r"mlp.fc1.\1"),
    (r"output\.dense\.(weight|bias)$", r"mlp.fc2.\1"),
]
_BUFFERS = ("relative_position_index", "attn_mask")


def canonical_encoder_keys(sd: Dict[str, object]) -> Dict[str, object]:
    """State dict of `encoder.molscribe_encoder` in any of three namings -> canonical `swin.*` keys (values untouched, except that a
    fused timm `attn.qkv` tensor is split into q / k / v):
      * timm 0.4.12 SwinTransformer (MolScribe: `patch_embed.proj`, `layers.i.blocks.j.attn.qkv`, `layers.i.downsample`, `norm`),
        behind any prefix (`cnn.`, `transformer.`, ...) - INFERRED to be what the fork's checkpoint holds;
      * stock transformers 5.x SwinModel (`embeddings.patch_embeddings.projection`, `attention.q_proj`, ...);
      * transformers 4.x SwinModel (`attention.self.query`, `attention.output.dense`, `intermediate.dense`, `output.dense`).
    Buffers (`relative_position_index`, `attn_mask`) and a classification head are dropped."""
    keys = list(sd.keys())
    timm_anchor = next((k for k in keys if k.endswith("patch_embed.proj.weight")), None)
    hf_anchor = next((k for k in keys if k.endswith("embeddings.patch_embeddings.projection.weight")), None)
    out: Dict[str, object] = {}
    if timm_anchor is not None:
        pre = timm_anchor[: -len("patch_embed.proj.weight")]
        for k, v in sd.items():
            if not k.startswith(pre):
                continue
            r = k[len(pre):]
            if any(b in r for b in _BUFFERS) or r.startswith("head."):
                continue
            m = re.match(r"patch_embed\.proj\.(weight|bias)$", r)
            if m:
                out[f"swin.embeddings.patch_embeddings.projection.{m.group(1)}"] = v
                continue
            m = re.match(r"patch_embed\.norm\.(weight|bias)$", r)
            if m:
                out[f"swin.embeddings.norm.{m.group(1)}"] = v
                continue
            m = re.match(r"norm\.(weight|bias)$", r)
            if m:
                out[f"swin.layernorm.{m.group(1)}"] = v
                continue
            m = re.match(r"layers\.(\d+)\.downsample\.(norm|reduction)\.(weight|bias)$", r)
            if m:
                out[f"swin.encoder.layers.{m.group(1)}.downsample.{m.group(2)}.{m.group(3)}"] = v
                continue
            m = re.match(r"layers\.(\d+)\.blocks\.(\d+)\.(.*)$", r)
            if not m:
                raise KeyError(f"e1 encoder: unrecognised timm key {k!r}")
            p = f"swin.encoder.layers.{m.group(1)}.blocks.{m.group(2)}."
            tail = m.group(3)
            q = re.match(r"attn\.qkv\.(weight|bias)$", tail)
            if q:
                n3 = v.shape[0]
                if n3 % 3:
                    raise ValueError(f"{k}: first dimension {n3} is not 3 x C")
                c = n3 // 3
                for i, nm in enumerate(("q_proj", "k_proj", "v_proj")):       # timm: qkv(x).reshape(.., 3, heads, hd) -> q, k, v in this order
                    out[p + f"attention.{nm}.{q.group(1)}"] = v[i * c:(i + 1) * c]
                continue
            for pat, rep in _TIMM_BLOCK:
                if re.match(pat, tail):
                    out[p + re.sub(pat, rep, tail)] = v
                    break
            else:
                raise KeyError(f"e1 encoder: unrecognised timm key {k!r}")
        return out
    if hf_anchor is None:
        raise KeyError("e1 encoder: neither a timm (`patch_embed.proj.weight`) nor a transformers (`embeddings.patch_embeddings."
                       "projection.weight`) Swin state dict")
    pre = hf_anchor[: -len("embeddings.patch_embeddings.projection.weight")]
    for k, v in sd.items():
        if not k.startswith(pre):
            continue
        r = k[len(pre):]
        if any(b in r for b in _BUFFERS) or r.startswith("pooler.") or r.startswith("classifier."):
            continue
        m = re.match(r"(encoder\.layers\.\d+\.blocks\.\d+\.)(.*)$", r)
        if m:
            tail = m.group(2)
            for pat, rep in _HF4_BLOCK:
                if re.match(pat, tail):
                    tail = re.sub(pat, rep, tail)
                    break
            r = m.group(1) + tail
        out["swin." + r] = v
    return out


def canonical_projector_keys(sd: Dict[str, object]) -> Dict[str, object]:
    """State dict of `encoder.molscribe_projector` -> `proj.{j}.weight|bias`, j = rank of the Linear among the module's 2-D weights in
    key order (an nn.Sequential(Linear, GELU, Linear) holds `0.*` and `2.*`; named layers such as `fc1` / `fc2` sort the same way).
    LayerNorm-like 1-D-only entries are rejected: the projector form is INFERRED as Linear (+ activation) layers."""
    groups: Dict[str, Dict[str, object]] = {}
    for k, v in sd.items():
        stem, _, leaf = k.rpartition(".")
        if leaf not in ("weight", "bias"):
            raise KeyError(f"e1 projector: unexpected entry {k!r}")
        groups.setdefault(stem, {})[leaf] = v

    def order(stem):
        return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", stem)]
    out: Dict[str, object] = {}
    for j, stem in enumerate(sorted(groups, key=order)):
        g = groups[stem]
        if "weight" not in g or len(g["weight"].shape) != 2:
            raise KeyError(f"e1 projector: {stem!r} is not a Linear layer (only Linear + activation stacks are supported)")
        out[f"proj.{j}.weight"] = g["weight"]
        if "bias" in g:
            out[f"proj.{j}.bias"] = g["bias"]
    return out


def shape_from_state(base: E1Shape, canon: Dict[str, object]) -> E1Shape:
    """`base` with the projector sizes read off the canonical state dict (proj.{j}.weight shapes)."""
    import dataclasses
    dims = []
    j = 0
    while f"proj.{j}.weight" in canon:
        dims.append(tuple(int(x) for x in canon[f"proj.{j}.weight"].shape))
        j += 1
    if not dims:
        return base
    if dims[0][1] != base.out_dim:
        raise ValueError(f"e1 projector: first Linear takes {dims[0][1]} features, the encoder produces {base.out_dim}")
    return dataclasses.replace(base, proj_dims=tuple(d[0] for d in dims[:-1]), d_model=dims[-1][0])
