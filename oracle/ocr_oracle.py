"""CPU oracle of the ChemicalOCR stage - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(markushgrapher_amd/ocr.py -> libmgrapher_hip.so) never does and fails loudly without its HIP library.

A plain torch-CPU fp32 restatement of stock `Idefics3ForConditionalGeneration` (transformers 5.15) as the reference's
ChemicalOCR calls it (markushgrapher/ocr/chemical_ocr.py:366-392: processor(...) then model.generate(**inputs,
max_new_tokens=4096, do_sample=False)):
  vision embeddings      models/idefics3/modeling_idefics3.py:98-172   (full-image case: position ids = arange)
  vision layer / tower   :199-356, :433-505    (LayerNorm eps, q/k/v/out with bias, softmax(q k^T / sqrt(64)), gelu_pytorch_tanh MLP)
  connector              :391-412              (pixel shuffle + bias-free projection)
  inputs_merger          :533-561              (image features replace the <image> token embeddings, in order)
  text model             models/llama/modeling_llama.py (RMSNorm, rotary "default", grouped-query causal attention, SwiGLU)
  greedy search          generation/utils.py:2783-2975 (argmax, finished rows emit pad, stop when every row has emitted EOS)
Pinned: tools/make_golden_ocr.py runs the stock class on the same weights and inputs in the build container and asserts
equality before it writes tests/golden/ocr_*.npz.  The reference's own checkpoint and its exact geometry are not available
offline: parity is pinned on the stock architecture, the checkpoint geometry is INFERRED.
"""
from __future__ import annotations


import torch


def _bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


class OcrOracle:
    def __init__(self, shape, state_dict, emulate_bf16: bool = False):
        self.s = shape
        self.w = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in state_dict.items()}
        self.emu = emulate_bf16
        if shape.tie_word_embeddings and "lm_head.weight" not in self.w:
            self.w["lm_head.weight"] = self.w["model.text_model.embed_tokens.weight"]

    def _r(self, x):                       # operand rounding of the HIP path (GEMM inputs, q/k/v)
        return _bf16(x) if self.emu else x

    def _lin(self, x, name, bias=True):
        y = self._r(x) @ self.w[name + ".weight"].T
        if bias and (name + ".bias") in self.w:
            y = y + self.w[name + ".bias"]
        return y

    # ---- vision tower (modeling_idefics3.py:98-172, 199-356, 433-505) -------------------------------------------------
    def patch_inputs(self, pixel_attention_mask):
        """pixel_attention_mask [N][I][I] bool -> (patch mask [N][P] bool, position ids [N][P] long): the patch grid of
        get_image_features (modeling_idefics3.py:605-608) and the fractional-coordinate buckets of Idefics3VisionEmbeddings
        (:128-172), evaluated with the same torch ops in fp32."""
        s = self.s
        ps, g = s.patch_size, s.image_size // s.patch_size
        pam = torch.as_tensor(pixel_attention_mask).bool()
        N = pam.shape[0]
        sub = pam.unfold(1, ps, ps).unfold(2, ps, ps)
        pmask = (sub.sum(dim=(-1, -2)) > 0)                                   # [N][g][g]
        boundaries = torch.arange(1 / g, 1.0, 1 / g)
        nb_h, nb_w = pmask[:, :, 0].sum(dim=1), pmask[:, 0, :].sum(dim=1)
        step_h, step_w = 1.0 / nb_h, 1.0 / nb_w
        idx = torch.arange(g, dtype=torch.float32)
        fh = torch.clamp(idx[None, :] * step_h[:, None], max=(1.0 - 1e-6)).to(torch.float32)
        fw = torch.clamp(idx[None, :] * step_w[:, None], max=(1.0 - 1e-6)).to(torch.float32)
        bh, bw = torch.bucketize(fh, boundaries, right=True), torch.bucketize(fw, boundaries, right=True)
        pos = (bh[:, :, None] * g + bw[:, None, :]).reshape(N, -1)
        flat = pmask.reshape(N, -1)
        position_ids = torch.zeros_like(pos)
        position_ids[flat] = pos[flat]
        return flat, position_ids

    def vision(self, pixel_values, pixel_attention_mask=None):
        """pixel_values [N][3][I][I] -> last_hidden_state [N][P][v_hidden] (after post_layernorm); pixel_attention_mask [N][I][I]
        (None = full frames): masked patches take position id 0 and are not attended as keys."""
        s, w = self.s, self.w
        N = pixel_values.shape[0]
        ps, g = s.patch_size, s.image_size // s.patch_size
        v = "model.vision_model."
        # Conv2d(k = stride = patch) as a matrix product over (c, dy, dx)
        x = pixel_values.reshape(N, 3, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(N, g * g, 3 * ps * ps)
        wk = w[v + "embeddings.patch_embedding.weight"].reshape(s.v_hidden, -1)
        h = self._r(x) @ wk.T + w[v + "embeddings.patch_embedding.bias"]
        kmask = None
        if pixel_attention_mask is None:
            h = h + w[v + "embeddings.position_embedding.weight"][None]      # full image: position ids = arange(P)
        else:
            pmask, pos_ids = self.patch_inputs(pixel_attention_mask)
            h = h + w[v + "embeddings.position_embedding.weight"][pos_ids]
            kmask = pmask[:, None, None, :]                                      # keys of masked patches are excluded (bidirectional mask)
        H = s.v_heads
        for i in range(s.v_layers):
            p = f"{v}encoder.layers.{i}."
            x = torch.nn.functional.layer_norm(h, (s.v_hidden,), w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], s.v_eps)
            q = self._r(self._lin(x, p + "self_attn.q_proj") * (64 ** -0.5)).reshape(N, -1, H, 64).transpose(1, 2)
            k = self._r(self._lin(x, p + "self_attn.k_proj")).reshape(N, -1, H, 64).transpose(1, 2)
            vv = self._r(self._lin(x, p + "self_attn.v_proj")).reshape(N, -1, H, 64).transpose(1, 2)
            sc = q @ k.transpose(-1, -2)
            if kmask is not None:
                sc = sc.masked_fill(~kmask, float("-inf"))
            a = torch.softmax(sc, dim=-1)
            ctx = (a @ vv).transpose(1, 2).reshape(N, -1, s.v_hidden)
            h = h + self._lin(ctx, p + "self_attn.out_proj")
            x = torch.nn.functional.layer_norm(h, (s.v_hidden,), w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], s.v_eps)
            y = torch.nn.functional.gelu(self._lin(x, p + "mlp.fc1"), approximate="tanh")
            h = h + self._lin(y, p + "mlp.fc2")
        return torch.nn.functional.layer_norm(h, (s.v_hidden,), w[v + "post_layernorm.weight"], w[v + "post_layernorm.bias"], s.v_eps)

    def pixel_shuffle(self, x):            # modeling_idefics3.py:397-406
        sf = self.s.scale_factor
        b, seq, e = x.shape
        hh = ww = int(seq ** 0.5)
        x = x.view(b, hh, ww, e).view(b, hh, ww // sf, e * sf).permute(0, 2, 1, 3)
        x = x.reshape(b, ww // sf, hh // sf, e * sf * sf).permute(0, 2, 1, 3)
        return x.reshape(b, seq // (sf * sf), e * sf * sf)

    def image_features(self, pixel_values, pixel_attention_mask=None):
        """pixel_values [B][n][3][I][I] (+ pixel_attention_mask [B][n][I][I]) -> [B*n][image_seq_len][t_hidden]
        (modeling_idefics3.py:563-622, all images real)."""
        pv = torch.as_tensor(pixel_values, dtype=torch.float32)
        pv = pv.reshape(-1, *pv.shape[2:])
        pam = None
        if pixel_attention_mask is not None:
            pam = torch.as_tensor(pixel_attention_mask).reshape(-1, *pv.shape[2:])
        return self._lin(self.pixel_shuffle(self.vision(pv, pam)), "model.connector.modality_projection.proj", bias=False)

    # ---- text model (modeling_llama.py) -------------------------------------------------------------------------------
    def _rope(self, x, pos):               # x [B][H][T][64], pos [T]
        inv = 1.0 / (self.s.rope_theta ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64))
        f = pos.to(torch.float32)[:, None] * inv[None]
        emb = torch.cat((f, f), dim=-1)
        cos, sin = emb.cos()[None, None], emb.sin()[None, None]
        rot = torch.cat((-x[..., 32:], x[..., :32]), dim=-1)
        return x * cos + rot * sin

    def _rms(self, h, name):
        var = h.pow(2).mean(-1, keepdim=True)
        return self.w[name] * (h * torch.rsqrt(var + self.s.rms_eps))

    def text(self, h, cache=None, pos0=0):
        """h [B][T][d] input embeddings at positions pos0.. -> final-norm hidden states; cache: list of (k, v) per layer."""
        s = self.s
        B, T, _ = h.shape
        H, KV = s.t_heads, s.t_kv_heads
        pos = torch.arange(pos0, pos0 + T)
        new_cache = []
        for i in range(s.t_layers):
            p = f"model.text_model.layers.{i}."
            x = self._rms(h, p + "input_layernorm.weight")
            q = self._lin(x, p + "self_attn.q_proj").reshape(B, T, H, 64).transpose(1, 2)
            k = self._lin(x, p + "self_attn.k_proj").reshape(B, T, KV, 64).transpose(1, 2)
            v = self._lin(x, p + "self_attn.v_proj").reshape(B, T, KV, 64).transpose(1, 2)
            q = self._r(self._rope(q, pos) * (64 ** -0.5))
            k = self._r(self._rope(k, pos))
            v = self._r(v)
            if cache is not None and cache[i] is not None:
                k = torch.cat((cache[i][0], k), dim=2)
                v = torch.cat((cache[i][1], v), dim=2)
            new_cache.append((k, v))
            kk = k.repeat_interleave(H // KV, dim=1)
            vv = v.repeat_interleave(H // KV, dim=1)
            sc = q @ kk.transpose(-1, -2)
            Tk = kk.shape[2]
            causal = torch.arange(Tk)[None, :] <= (pos[:, None])
            sc = sc.masked_fill(~causal[None, None], float("-inf"))
            ctx = (torch.softmax(sc, dim=-1) @ vv).transpose(1, 2).reshape(B, T, H * 64)
            h = h + self._lin(ctx, p + "self_attn.o_proj")
            x = self._rms(h, p + "post_attention_layernorm.weight")
            y = torch.nn.functional.silu(self._lin(x, p + "mlp.gate_proj")) * self._lin(x, p + "mlp.up_proj")
            h = h + self._lin(y, p + "mlp.down_proj")
        return self._rms(h, "model.text_model.norm.weight"), new_cache

    def embed(self, input_ids, pixel_values, pixel_attention_mask=None):
        ids = torch.as_tensor(input_ids, dtype=torch.long)
        h = self.w["model.text_model.embed_tokens.weight"][ids]
        if pixel_values is not None:
            feats = self.image_features(pixel_values, pixel_attention_mask)
            m = ids == self.s.image_token_id
            h = h.clone()
            h[m] = feats.reshape(-1, feats.shape[-1])          # masked_scatter: row-major order of the <image> positions
        return h

    def forward(self, input_ids, pixel_values, pixel_attention_mask=None):
        """Teacher-forced logits [B][L][V] (Idefics3ForConditionalGeneration.forward, modeling_idefics3.py:750-840)."""
        hs, _ = self.text(self.embed(input_ids, pixel_values, pixel_attention_mask))
        return self._r(hs) @ self.w["lm_head.weight"].T

    def generate_padded(self, input_ids, attention_mask, pixel_values, max_new_tokens, return_logits=False, pixel_attention_mask=None):
        """A LEFT-PADDED batch as the Idefics3 processor makes one from prompts of different lengths (padding_side = "left"): attention_mask
        [B][L] = 0 over the leading pad tokens.  Stock gives every real token the position cumsum(mask) - 1 (generation/utils.py
        `prepare_inputs_for_generation`, modeling_llama.py position_ids) and masks the pad keys (masking_utils / _update_causal_mask), so a
        row's result is the result of that row ALONE without its padding; restated exactly so: each row through generate() on its own.
        Returns new ids [B][n] (rows padded with pad_token_id to the longest) and, optionally, the step logits [B][n][V]."""
        ids = torch.as_tensor(input_ids, dtype=torch.long)
        am = torch.as_tensor(attention_mask).bool()
        B = ids.shape[0]
        news, logs = [], []
        for b in range(B):
            p = int((~am[b]).sum())
            assert bool(am[b, p:].all()) and not bool(am[b, :p].any()), "left padding: zeros in front, ones behind"
            pv = None if pixel_values is None else torch.as_tensor(pixel_values)[b:b + 1]
            pam = None if pixel_attention_mask is None else torch.as_tensor(pixel_attention_mask)[b:b + 1]
            r = self.generate(ids[b:b + 1, p:], pv, max_new_tokens, return_logits=True, pixel_attention_mask=pam)
            news.append(r[0][0]); logs.append(r[1][0])
        n = max(int(x.shape[0]) for x in news)
        new = torch.full((B, n), self.s.pad_token_id, dtype=torch.long)
        lg = torch.zeros((B, n, logs[0].shape[-1]))
        for b in range(B):
            new[b, :news[b].shape[0]] = news[b]
            lg[b, :logs[b].shape[0]] = logs[b]
        return (new, lg) if return_logits else new

    def generate(self, input_ids, pixel_values, max_new_tokens, return_logits=False, pixel_attention_mask=None):
        """Greedy search (generation/utils.py:2783-2975): returns new ids [B][n <= max_new_tokens] (pad after EOS)."""
        s = self.s
        ids = torch.as_tensor(input_ids, dtype=torch.long)
        B, L = ids.shape
        hs, cache = self.text(self.embed(ids, pixel_values, pixel_attention_mask))
        logits = self._r(hs[:, -1]) @ self.w["lm_head.weight"].T
        unfinished = torch.ones(B, dtype=torch.bool)
        out, step_logits = [], []
        for t in range(max_new_tokens):
            step_logits.append(logits)
            nxt = logits.argmax(-1)
            nxt = torch.where(unfinished, nxt, torch.full_like(nxt, s.pad_token_id))
            out.append(nxt)
            for e in (s.eos_token_id,) + tuple(getattr(s, "eos_extra", ())):      # a list of EOS ids stops on any of them (gen:2927-2937)
                unfinished = unfinished & (nxt != e)
            if not bool(unfinished.any()) or t + 1 == max_new_tokens:
                break
            h = self.w["model.text_model.embed_tokens.weight"][nxt][:, None]
            hs, cache = self.text(h, cache, pos0=L + t)
            logits = self._r(hs[:, -1]) @ self.w["lm_head.weight"].T
        new = torch.stack(out, dim=1)
        return (new, torch.stack(step_logits, dim=1)) if return_logits else new
