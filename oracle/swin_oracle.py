"""CPU oracle of the OCSR vision branch "e1"  - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(markushgrapher_amd/e1.py -> libmgrapher_hip.so) never does and fails loudly without its HIP library.

A plain torch-CPU fp32 restatement of stock `transformers.SwinModel` (transformers 5.15, models/swin/modeling_swin.py), which is
the importable upstream of MolScribe's timm 0.4.12 `swin_base_patch4_window12_384` encoder that the reference loads into
`model.encoder.molscribe_encoder` (ref: markushgrapher/core/common/begin.py:137-138, utils/model/utils_model_loading.py:20-36,
README.md:212-215):
  patch embedding + LayerNorm     modeling_swin.py:219-245, 277-286   (Conv2d k = stride = patch as a matrix product; no absolute positions)
  window partition / reverse      :486-505
  cyclic shift + region mask      :584-607, 617-626                   (mask value -100 between different regions, NOT -inf)
  relative position bias          :339-370                            (table [(2w-1)^2, heads], index (dy + w - 1)(2w - 1) + dx + w - 1)
  window attention                :373-398, 418-468                   (softmax(q k^T / sqrt(32) + bias + mask) v, q/k/v/o with bias)
  block                           :529-574                            (pre-LN, exact-erf GELU MLP, window clamped to the map: :576-582)
  patch merging                   :309-326                            (x[0::2,0::2] | x[1::2,0::2] | x[0::2,1::2] | x[1::2,1::2] -> LN -> Linear)
  final LayerNorm                 :885-887                            (`last_hidden_state`; no pooler)
Pinned: tools/make_golden_swin.py runs the stock class on the same weights and inputs in the build container and asserts equality
before it writes tests/golden/swin_*.npz.

PARITY UNPINNED (the fork's `modeling_markushgrapher.py` is not in the reference tree) - restated here from the
README's description only and marked INFERRED in markushgrapher_amd/e1_shapes.py: `derive_input` (bilinear 512 -> 384 resize of the
model's pixel_values, optional re-normalisation), `project` (Linear / GELU stack), and the concatenation [e1 | e2] in front of the decoder
(udop_oracle.Oracle.fuse_e1).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


class SwinOracle:
    def __init__(self, shape, state_dict, emulate_bf16: bool = False):
        """state_dict: canonical keys of markushgrapher_amd.e1_shapes.state_dict_spec (`swin.*`, `proj.*`).  emulate_bf16: round where the
        HIP path stores bf16 (GEMM operands, q / k / v, softmax weights, attention output, GELU output)."""
        self.s = shape
        self.w = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in state_dict.items()}
        self.emu = emulate_bf16

    def _r(self, x):
        return _bf16(x) if self.emu else x

    def _lin(self, x, name):
        y = self._r(x) @ self.w[name + ".weight"].T
        if (name + ".bias") in self.w:
            y = y + self.w[name + ".bias"]
        return y

    def _ln(self, x, name, eps=None):
        """eps None = config.layer_norm_eps (the blocks' and the final norm: stock:520-521, 837); embeddings.norm and the merging norm
        are plain nn.LayerNorm(dim) with torch's default 1e-5 (stock:183, 301)."""
        return F.layer_norm(x, (x.shape[-1],), self.w[name + ".weight"], self.w[name + ".bias"], self.s.layer_norm_eps if eps is None else eps)

    # ---- INFERRED input derivation ------------------------------------------------------------------------------------------
    def derive_input(self, pixel_values):
        """[B, 3, src, src] -> [B, 3, I, I]: F.interpolate(bilinear, align_corners=False, no antialias), then x * scale_c + shift_c."""
        s = self.s
        x = torch.as_tensor(pixel_values, dtype=torch.float32)
        if x.shape[-1] != s.image_size or x.shape[-2] != s.image_size:
            x = F.interpolate(x, size=(s.image_size, s.image_size), mode="bilinear", align_corners=False)
        sc = torch.tensor(s.pix_scale, dtype=torch.float32).view(1, 3, 1, 1)
        sh = torch.tensor(s.pix_shift, dtype=torch.float32).view(1, 3, 1, 1)
        return x * sc + sh

    # ---- stock SwinModel ------------------------------------------------------------------------------------------------------
    def _rel_index(self, w):
        """modeling_swin.py:350-365 for a w x w window against a table laid out for the CONFIGURED window W: the stock module builds
        the index for W and, when the window is clamped (w < W), the attention still adds the W-sized bias - which only type-checks
        when w == W.  Stock therefore requires w == W wherever attention runs with a bias of matching size; here w == W or the
        stage's map equals the window (clamped case handled by the caller)."""
        c = torch.arange(w)
        cy, cx = torch.meshgrid(c, c, indexing="ij")
        cy, cx = cy.reshape(-1), cx.reshape(-1)
        dy = cy[:, None] - cy[None, :] + (w - 1)
        dx = cx[:, None] - cx[None, :] + (w - 1)
        return dy * (2 * w - 1) + dx

    def _block(self, h, i, j, R):
        s = self.s
        C, H = s.stage_dim(i), s.num_heads[i]
        B = h.shape[0]
        w = s.stage_window(i)
        if w != s.window_size:
            # stock adds a (W^2 x W^2) bias to (w^2 x w^2) scores: shapes only agree for w == W (modeling_swin.py:435-448)
            raise ValueError(f"stage {i}: map {R} smaller than window {s.window_size} is not representable in stock SwinModel")
        shift = 0 if (j % 2 == 0 or R <= s.window_size) else s.window_size // 2
        if R % w:
            raise ValueError("maps that need padding to a multiple of the window are outside this oracle (the reference geometry has none)")
        p = f"swin.encoder.layers.{i}.blocks.{j}."
        x = self._ln(h, p + "layernorm_before").view(B, R, R, C)
        if shift:
            x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
        nw = R // w
        xw = x.view(B, nw, w, nw, w, C).transpose(2, 3).reshape(B * nw * nw, w * w, C)
        q = self._r(self._lin(xw, p + "attention.q_proj")).view(-1, w * w, H, 32).transpose(1, 2)
        k = self._r(self._lin(xw, p + "attention.k_proj")).view(-1, w * w, H, 32).transpose(1, 2)
        v = self._r(self._lin(xw, p + "attention.v_proj")).view(-1, w * w, H, 32).transpose(1, 2)
        tab = self.w[p + "attention.relative_position_bias.relative_position_bias_table"]
        bias = tab[self._rel_index(w).reshape(-1)].view(w * w, w * w, H).permute(2, 0, 1)
        sc = (q @ k.transpose(2, 3)) * (32 ** -0.5) + bias[None]
        if shift:
            idx = torch.arange(R)
            reg = (idx >= R - w).long() + (idx >= R - shift).long()
            img = (reg[:, None] * 3 + reg[None, :]).view(nw, w, nw, w).transpose(1, 2).reshape(nw * nw, w * w)
            m = (img[:, None, :] != img[:, :, None]).to(torch.float32) * -100.0           # [windows][q][k]
            sc = (sc.view(B, nw * nw, H, w * w, w * w) + m[None, :, None]).view(-1, H, w * w, w * w)
        if self.emu:
            pr = _bf16(torch.exp(sc - sc.max(dim=-1, keepdim=True).values))
            ctx = (pr @ v) / pr.sum(dim=-1, keepdim=True)
        else:
            ctx = torch.softmax(sc, dim=-1) @ v
        ctx = ctx.transpose(1, 2).reshape(-1, w * w, C)
        a = self._lin(ctx, p + "attention.o_proj")
        a = a.view(B, nw, nw, w, w, C).transpose(2, 3).reshape(B, R, R, C)
        if shift:
            a = torch.roll(a, shifts=(shift, shift), dims=(1, 2))
        h = h + a.reshape(B, R * R, C)
        y = self._lin(self._ln(h, p + "layernorm_after"), p + "mlp.fc1")
        y = F.gelu(y)                                                   # hidden_act "gelu" = exact erf form
        return h + self._lin(y, p + "mlp.fc2")

    def features(self, pixels):
        """pixels [B, 3, I, I] (the branch's own input) -> last_hidden_state [B, M, C_last]."""
        s = self.s
        x = torch.as_tensor(pixels, dtype=torch.float32)
        B = x.shape[0]
        ps, g = s.patch_size, s.grid
        cols = x.reshape(B, s.num_channels, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, -1)
        wk = self.w["swin.embeddings.patch_embeddings.projection.weight"].reshape(s.embed_dim, -1)
        h = self._r(cols) @ wk.T + self.w["swin.embeddings.patch_embeddings.projection.bias"]
        h = self._ln(h, "swin.embeddings.norm", 1e-5)
        for i in range(s.n_stages):
            R, C = s.stage_res(i), s.stage_dim(i)
            for j in range(s.depths[i]):
                h = self._block(h, i, j, R)
            if i + 1 < s.n_stages:
                p = f"swin.encoder.layers.{i}.downsample."
                x4 = h.view(B, R, R, C)
                x4 = torch.cat([x4[:, r::2, c::2, :] for c in range(2) for r in range(2)], dim=-1).reshape(B, -1, 4 * C)
                h = self._r(self._ln(x4, p + "norm", 1e-5)) @ self.w[p + "reduction.weight"].T
        return self._ln(h, "swin.layernorm")

    # ---- INFERRED projector ---------------------------------------------------------------------------------------------------
    def project(self, feats):
        s = self.s
        n = len(s.proj_dims) + 1
        y = feats
        for j in range(n):
            y = self._lin(y, f"proj.{j}")
            if j + 1 < n and s.proj_act == "gelu":
                y = F.gelu(y)
        return y

    def e1(self, pixel_values):
        """The whole branch: the VTL model's pixel_values [B, 3, src, src] -> e1 [B, M, d_model]."""
        return self.project(self.features(self.derive_input(pixel_values)))

    # ---- bookkeeping for the benchmark -----------------------------------------------------------------------------------------
    @staticmethod
    def flops_per_image(s) -> float:
        """Algorithmic flops of one image through encoder + projector (2 x multiply-accumulates; attention included)."""
        g = s.grid
        f = 2.0 * g * g * s.embed_dim * s.num_channels * s.patch_size ** 2
        for i in range(s.n_stages):
            R, C = s.stage_res(i), s.stage_dim(i)
            w = s.stage_window(i)
            f += s.depths[i] * (R * R * 2.0 * C * C * (4 + 2 * s.mlp_ratio) + R * R * 4.0 * w * w * C)
            if i + 1 < s.n_stages:
                f += (R * R / 4) * 2.0 * 4 * C * 2 * C
        dims = (s.out_dim,) + tuple(s.proj_dims) + (s.d_model,)
        for a, b in zip(dims[:-1], dims[1:]):
            f += s.out_tokens * 2.0 * a * b
        return f
