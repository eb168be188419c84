"""CPU ORACLE — test infrastructure only.  NOT part of the product path.

A plain fp32 (torch-CPU) restatement of the arithmetic the reference delegates to its transformers fork
(`transformers.models.markushgrapher`, un-vendored and unpinned — SURVEY.md §0).  The importable upstream
of that arithmetic is stock transformers 5.15.0 `models/udop/modeling_udop.py` ("stock:" below) and
`generation/utils.py` ("gen:" below); each function cites the lines it follows.

Pinned by: tests/golden/*.npz, minted in the build container from stock UdopForConditionalGeneration by
tools/make_golden.py (the stock model itself never travels).  The reference itself has NO golden vectors
or tests for this path, and the fork-only pieces (Swin-B e1 branch, projector) are
"parity unpinned" (SURVEY.md §8c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

`emulate_bf16=True` inserts bf16 round-trips at the points where the HIP path stores bf16 (see DESIGN.md
"Precision map"); it is used to bound the HIP path's expected deviation, not as a reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

F32 = torch.float32


def _t(x):
    if isinstance(x, torch.Tensor):
        return x
    return torch.from_numpy(np.ascontiguousarray(x))


def _bf(x: torch.Tensor, on: bool) -> torch.Tensor:
    return x.to(torch.bfloat16).to(F32) if on else x


# ----------------------------------------------------------------------------------------------
# relative-position buckets  (stock:422-468)
# ----------------------------------------------------------------------------------------------
def relative_position_bucket(rel: torch.Tensor, bidirectional: bool, num_buckets: int, max_distance: int):
    """stock:422-468 `_relative_position_bucket` restated."""
    rel = rel.to(torch.long)
    buckets = torch.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        buckets = buckets + (rel > 0).to(torch.long) * num_buckets
        rel = rel.abs()
    else:
        rel = -torch.minimum(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    large = max_exact + (
        torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)
    ).to(torch.long)
    large = torch.minimum(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rel, large)


def bucket_table(bidirectional: bool, num_buckets: int, max_distance: int, lo: int, hi: int) -> np.ndarray:
    """bucket id for every integer distance in [lo, hi] (used to pin the HIP library's host-built tables)."""
    rel = torch.arange(lo, hi + 1, dtype=torch.long)
    return relative_position_bucket(rel, bidirectional, num_buckets, max_distance).numpy().astype(np.int32)


class Oracle:
    def __init__(self, shape, state_dict: Dict[str, np.ndarray], emulate_bf16: bool = False):
        self.s = shape
        self.w = {k: _t(v).to(F32) for k, v in state_dict.items()}
        self.bf = emulate_bf16
        self.threads = torch.get_num_threads()

    # ------------------------------------------------------------------------------------------
    def rmsnorm(self, x, g):
        """stock:293-306 UdopLayerNorm (fp32 variance, eps 1e-6, no mean, no bias)."""
        var = x.pow(2).mean(-1, keepdim=True)
        return g * (x * torch.rsqrt(var + self.s.layer_norm_epsilon))

    def linear(self, x, key):
        return x @ self.w[key].t()

    # ------------------------------------------------------------------------------------------
    def patch_embed(self, pixel_values):
        """stock:254-280 Conv2d(k=16,s=16)+bias, flatten(2).transpose(1,2) → [B,P,d]."""
        s = self.s
        B, C, Hh, Ww = pixel_values.shape
        ps = s.patch_size
        n = Hh // ps
        x = pixel_values.reshape(B, C, n, ps, n, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, n * n, C * ps * ps)
        x = _bf(x, self.bf)
        wgt = self.w["patch_embed.proj.weight"].reshape(s.d_model, -1)
        return x @ wgt.t() + self.w["patch_embed.proj.bias"]

    def visual_bbox(self):
        """stock:135-155 get_visual_bbox (fp32 arithmetic)."""
        n = self.s.image_size // self.s.patch_size
        xs = torch.arange(0, 1.0 * (n + 1), 1.0) / n
        x0 = xs[:-1].repeat(n, 1)
        y0 = xs[:-1].repeat(n, 1).transpose(0, 1)
        x1 = xs[1:].repeat(n, 1)
        y1 = xs[1:].repeat(n, 1).transpose(0, 1)
        return torch.stack([x0, y0, x1, y1], dim=-1).reshape(-1, 4)

    def combine(self, image_emb, tok_emb, bbox, attention_mask):
        """stock:171-251 combine_image_text_embeddings, incl. the "drop for every token" quirk
        (SURVEY.md §8 a4).  bbox fp32 in; returns (embeds [B,S,d], bbox f64 [B,S,4], mask or None)."""
        s = self.s
        n = s.image_size // s.patch_size
        B, L, _ = tok_emb.shape
        P = image_emb.shape[1]
        px = torch.clip(torch.floor((bbox[:, :, 0] + bbox[:, :, 2]) / 2.0 * n).long(), 0, n - 1)
        py = torch.clip(torch.floor((bbox[:, :, 1] + bbox[:, :, 3]) / 2.0 * n).long(), 0, n - 1) * n
        pts = px + py
        bbox64 = bbox.to(torch.float64)
        mean = bbox64.mean(-1)
        target_seg = (mean == 0.0) | (mean == 1.0)
        rep = torch.gather(image_emb, 1, pts.unsqueeze(-1).repeat(1, 1, image_emb.size(-1)))
        rep = rep.clone()
        rep[target_seg] = 0.0
        tok_emb = tok_emb + rep
        keep = torch.ones(B, P, dtype=torch.bool)
        for b in range(B):
            keep[b, pts[b]] = False
        vb = self.visual_bbox().to(torch.float64)
        out_e = torch.zeros(B, L + P, s.d_model, dtype=F32)
        out_b = torch.zeros(B, L + P, 4, dtype=torch.float64)
        out_m = None if attention_mask is None else torch.zeros(B, L + P, dtype=torch.long)
        for b in range(B):
            k = int(keep[b].sum())
            out_e[b, :L] = tok_emb[b]
            out_e[b, L:L + k] = image_emb[b][keep[b]]
            out_b[b, :L] = bbox64[b]
            out_b[b, L:L + k] = vb[keep[b]]
            if out_m is not None:
                out_m[b, :L] = attention_mask[b]
                out_m[b, L:L + k] = 1
        return out_e, out_b, out_m

    def cell_embed(self, bbox64):
        """stock:822-840 UdopCellEmbeddings (bbox is float64 after combine — stock:200)."""
        m = self.s.max_2d_position_embeddings
        bb = torch.clip(bbox64, 0.0, 1.0)
        idx = (bb * (m - 1)).long().clamp(0, m - 1)
        xe = self.w["encoder.cell_2d_embedding.x_position_embeddings.weight"]
        ye = self.w["encoder.cell_2d_embedding.y_position_embeddings.weight"]
        return xe[idx[:, :, 0]] + ye[idx[:, :, 1]] + xe[idx[:, :, 2]] + ye[idx[:, :, 3]], idx

    def encoder_bias(self, bbox64, S):
        """stock:904-953,956-1029: sum of 1-D, horizontal and vertical bucketed biases → [B,H,S,S]."""
        s = self.s
        nb = s.relative_attention_num_buckets
        pos = torch.arange(S, dtype=torch.long)[None, :]
        rel1 = (pos[:, None, :] - pos[:, :, None]).float().to(torch.long)
        b1 = relative_position_bucket(rel1, True, nb, 128)
        cx = bbox64[:, :, [0, 2]].mean(dim=-1)
        cy = bbox64[:, :, [1, 3]].mean(dim=-1)

        def rel2(c):
            r = (c[:, None, :] - c[:, :, None]).float()
            r = r * 100
            return r.to(torch.long)
        bh = relative_position_bucket(rel2(cx), True, nb, 100)
        bv = relative_position_bucket(rel2(cy), True, nb, 100)
        t1 = self.w["encoder.relative_bias.biases.0.relative_attention_bias.weight"]
        th = self.w["encoder.relative_bias.biases.1.relative_attention_bias.weight"]
        tv = self.w["encoder.relative_bias.biases.2.relative_attention_bias.weight"]
        bias = t1[b1].permute(0, 3, 1, 2) + th[bh].permute(0, 3, 1, 2) + tv[bv].permute(0, 3, 1, 2)
        return bias, (b1, bh, bv)

    def attention(self, q, k, v, bias, keymask_add):
        """stock:59-87 eager attention; scaling = 1.0 (stock:402-403); q,k,v [B,H,T,dk]."""
        scores = q @ k.transpose(2, 3)
        if bias is not None:
            scores = scores + bias
        if keymask_add is not None:
            scores = scores + keymask_add
        if self.bf:
            mx = scores.max(dim=-1, keepdim=True).values
            p = _bf(torch.exp(scores - mx), True)
            out = (p @ v) / p.sum(dim=-1, keepdim=True)
        else:
            p = torch.softmax(scores, dim=-1)
            out = p @ v
        return out

    def _heads(self, x):
        B, T, _ = x.shape
        return x.view(B, T, self.s.num_heads, self.s.d_kv).transpose(1, 2)

    def _merge(self, x):
        B, H, T, dk = x.shape
        return x.transpose(1, 2).reshape(B, T, H * dk)

    # ------------------------------------------------------------------------------------------
    def encode(self, input_ids, bbox, pixel_values, attention_mask=None, return_parts=False):
        """stock:1102-1246 UdopStack.forward (encoder).  Returns (enc_out [B,S,d] f32, mask [B,S] i64)."""
        s = self.s
        input_ids = _t(input_ids).long()
        bbox = _t(bbox).to(F32)
        pixel_values = _t(pixel_values).to(F32)
        am = None if attention_mask is None else _t(attention_mask).long()
        tok = self.w["shared.weight"][input_ids]
        img = self.patch_embed(pixel_values)
        emb, bbox64, mask = self.combine(img, tok, bbox, am)
        cell, cell_idx = self.cell_embed(bbox64)
        h = emb + cell
        B, S, _ = h.shape
        if mask is None:
            mask = torch.ones(B, S, dtype=torch.long)      # stock:1183-1186
        keymask = torch.zeros(B, 1, 1, S, dtype=F32)
        keymask.masked_fill_(mask[:, None, None, :] == 0, torch.finfo(F32).min)
        bias, buckets = self.encoder_bias(bbox64, S)
        parts = {"embed": h.clone(), "patch_emb": img, "cell_idx": cell_idx, "buckets": buckets, "bbox64": bbox64}
        def norm_in(hh, gain, deferred):
            # emulate_bf16 only: where the HIP encoder stores bf16.  Layer 0 normalises explicitly, bf16(RMSNorm(h)*g);
            # every later sub-layer input is the deferred form of DESIGN.md "Precision map": bf16(h*g) is what is stored
            # and the consuming projection scales its fp32 output rows by r = rsqrt(mean h^2 + eps) - by linearity the
            # same as feeding bf16(h*g)*r unrounded.  In fp32 mode both are the reference's RMSNorm (stock:293-306).
            if not (self.bf and deferred):
                return _bf(self.rmsnorm(hh, gain), self.bf)
            r = torch.rsqrt(hh.pow(2).mean(-1, keepdim=True) + self.s.layer_norm_epsilon)
            return _bf(hh * gain, True) * r
        for i in range(s.num_layers):
            p = f"encoder.block.{i}.layer"
            x = norm_in(h, self.w[f"{p}.0.layer_norm.weight"], i > 0)
            q = _bf(self._heads(self.linear(x, f"{p}.0.SelfAttention.q.weight")), self.bf)
            k = _bf(self._heads(self.linear(x, f"{p}.0.SelfAttention.k.weight")), self.bf)
            v = _bf(self._heads(self.linear(x, f"{p}.0.SelfAttention.v.weight")), self.bf)
            ctx = _bf(self._merge(self.attention(q, k, v, bias, keymask)), self.bf)
            h = h + self.linear(ctx, f"{p}.0.SelfAttention.o.weight")
            x = norm_in(h, self.w[f"{p}.1.layer_norm.weight"], True)
            y = _bf(torch.relu(self.linear(x, f"{p}.1.DenseReluDense.wi.weight")), self.bf)
            h = h + self.linear(y, f"{p}.1.DenseReluDense.wo.weight")
            if return_parts and i == 0:
                parts["h_layer0"] = h.clone()
        out = self.rmsnorm(h, self.w["encoder.final_layer_norm.weight"])
        if return_parts:
            return out, mask, parts
        return out, mask

    # ------------------------------------------------------------------------------------------
    def decoder_bias(self, q_pos: torch.Tensor, k_len: int):
        """stock:470-485 compute_bias for the decoder (bidirectional=False), table of block 0 shared by all
        layers (stock:1234-1237).  q_pos [Tq] absolute positions → [1,H,Tq,k_len]."""
        s = self.s
        mem = torch.arange(k_len, dtype=torch.long)[None, :]
        rel = mem - q_pos[:, None]
        b = relative_position_bucket(rel, False, s.relative_attention_num_buckets, s.relative_attention_max_distance)
        tab = self.w["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
        return tab[b].permute(2, 0, 1).unsqueeze(0)

    def cross_kv(self, enc_out):
        """stock:524-538 (first decoder step): per layer K_x = enc·Wk^T, V_x = enc·Wv^T."""
        s = self.s
        e = _bf(enc_out, self.bf)
        out = []
        for i in range(s.num_decoder_layers):
            p = f"decoder.block.{i}.layer.1.EncDecAttention"
            out.append((_bf(self._heads(self.linear(e, f"{p}.k.weight")), self.bf),
                        _bf(self._heads(self.linear(e, f"{p}.v.weight")), self.bf)))
        return out

    def decoder_stack(self, dec_ids, enc_mask, xkv, self_kv=None, past=0, dec_mask=None):
        """stock:1102-1246 UdopStack.forward (decoder) for T new positions starting at `past`.
        self_kv: list of (k,v) [B,H,past,dk] or None.  Returns (hidden [B,T,d] after final norm, new self_kv)."""
        s = self.s
        B, T = dec_ids.shape
        h = self.w["shared.weight"][dec_ids]
        total = past + T
        q_pos = torch.arange(past, total, dtype=torch.long)
        bias = self.decoder_bias(q_pos, total)
        causal = torch.zeros(1, 1, T, total, dtype=F32)
        kk = torch.arange(total)[None, :]
        causal.masked_fill_((kk > q_pos[:, None])[None, None], torch.finfo(F32).min)
        if dec_mask is not None:
            causal = causal.expand(B, 1, T, total).clone()
            causal.masked_fill_(_t(dec_mask)[:, None, None, :total] == 0, torch.finfo(F32).min)
        xmask = torch.zeros(B, 1, 1, enc_mask.shape[1], dtype=F32)
        xmask.masked_fill_(enc_mask[:, None, None, :] == 0, torch.finfo(F32).min)
        new_kv = []
        for i in range(s.num_decoder_layers):
            p = f"decoder.block.{i}.layer"
            x = _bf(self.rmsnorm(h, self.w[f"{p}.0.layer_norm.weight"]), self.bf)
            q = _bf(self._heads(self.linear(x, f"{p}.0.SelfAttention.q.weight")), self.bf)
            k = _bf(self._heads(self.linear(x, f"{p}.0.SelfAttention.k.weight")), self.bf)
            v = _bf(self._heads(self.linear(x, f"{p}.0.SelfAttention.v.weight")), self.bf)
            if self_kv is not None:
                k = torch.cat([self_kv[i][0], k], dim=2)
                v = torch.cat([self_kv[i][1], v], dim=2)
            new_kv.append((k, v))
            ctx = _bf(self._merge(self.attention(q, k, v, bias, causal)), self.bf)
            h = h + self.linear(ctx, f"{p}.0.SelfAttention.o.weight")
            x = _bf(self.rmsnorm(h, self.w[f"{p}.1.layer_norm.weight"]), self.bf)
            q = _bf(self._heads(self.linear(x, f"{p}.1.EncDecAttention.q.weight")), self.bf)
            ctx = _bf(self._merge(self.attention(q, xkv[i][0], xkv[i][1], None, xmask)), self.bf)
            h = h + self.linear(ctx, f"{p}.1.EncDecAttention.o.weight")
            x = _bf(self.rmsnorm(h, self.w[f"{p}.2.layer_norm.weight"]), self.bf)
            y = _bf(torch.relu(self.linear(x, f"{p}.2.DenseReluDense.wi.weight")), self.bf)
            h = h + self.linear(y, f"{p}.2.DenseReluDense.wo.weight")
        return self.rmsnorm(h, self.w["decoder.final_layer_norm.weight"]), new_kv

    def lm_logits(self, hidden):
        """stock:1554-1557: ×d^-0.5 (tied embeddings) then lm_head (= shared)."""
        x = _bf(hidden * (self.s.d_model ** -0.5), self.bf)
        return x @ self.w["shared.weight"].t()

    def default_generation_mask(self, input_ids):
        """gen:775-807 `_prepare_attention_mask_for_generation`: generate() without an attention_mask infers
        it from pad tokens when any are present (pad != eos), else all ones.  The reference calls generate with
        the mask deleted (ref: utils/ocsr/utils_evaluation.py:172-175) at batch size 1, i.e. all ones."""
        pad, eos = self.s.pad_token_id, self.s.eos_token_id
        if pad is not None and pad != eos and bool((input_ids == pad).any()):
            return (input_ids != pad).long()
        return torch.ones_like(input_ids)

    @staticmethod
    def shift_right(labels, start_id, pad_id):
        """stock:791-811 _shift_right."""
        labels = _t(labels).long()
        out = torch.zeros_like(labels)
        out[..., 1:] = labels[..., :-1]
        out[..., 0] = start_id
        out.masked_fill_(out == -100, pad_id)
        return out

    @staticmethod
    def fuse_e1(enc, mask, e1):
        """MarkushGrapher-2 late fusion (ref: README.md:212-215 "the projected vision embedding (e1) is concatenated with the VTL
        embedding (e2) and fed to a text decoder"): the decoder cross-attends over [e1 | e2], e1 tokens always attended.
        INFERRED - the fork's source is unavailable: this is the build's own statement, parity unpinned."""
        if e1 is None:
            return enc, mask
        e1 = _t(e1).to(F32)
        return torch.cat([e1, enc], dim=1), torch.cat([torch.ones(e1.shape[:2], dtype=mask.dtype), mask], dim=1)

    def forward(self, input_ids, bbox, pixel_values, attention_mask=None, labels=None, decoder_input_ids=None,
                decoder_attention_mask=None, e1=None):
        """stock:1448-1574 UdopForConditionalGeneration.forward → logits [B,T,V]."""
        enc, mask = self.encode(input_ids, bbox, pixel_values, attention_mask)
        enc, mask = self.fuse_e1(enc, mask, e1)
        if decoder_input_ids is None:
            decoder_input_ids = self.shift_right(labels, self.s.decoder_start_token_id, self.s.pad_token_id)
        dec_ids = _t(decoder_input_ids).long()
        hid, _ = self.decoder_stack(dec_ids, mask, self.cross_kv(enc), None, 0, decoder_attention_mask)
        return self.lm_logits(hid)

    # ------------------------------------------------------------------------------------------
    def greedy(self, input_ids, bbox, pixel_values, attention_mask=None, max_length=512, min_length=0,
               record=None, e1=None):
        """gen:2783-2975 `_sample` with do_sample=False: argmax, EOS/pad bookkeeping, max_length counts the
        start token.  HF's generate() builds an all-ones text mask when none is given (gen:
        `_prepare_attention_mask_for_generation`).  `min_length` mirrors MinLengthLogitsProcessor
        (EOS logit = -inf while cur_len < min_length).  Returns ids [B,T'] (i64 numpy)."""
        s = self.s
        input_ids = _t(input_ids).long()
        if attention_mask is None:
            attention_mask = self.default_generation_mask(input_ids)
        enc, mask = self.encode(input_ids, bbox, pixel_values, attention_mask)
        enc, mask = self.fuse_e1(enc, mask, e1)
        xkv = self.cross_kv(enc)
        B = input_ids.shape[0]
        seq = torch.full((B, 1), s.decoder_start_token_id, dtype=torch.long)
        unfinished = torch.ones(B, dtype=torch.long)
        kv = None
        cur = seq
        while True:
            hid, kv = self.decoder_stack(cur, mask, xkv, kv, seq.shape[1] - 1)
            logits = self.lm_logits(hid[:, -1:, :])[:, 0, :]
            if min_length and seq.shape[1] < min_length:
                logits[:, s.eos_token_id] = -float("inf")
            if record is not None:
                record.append(logits.clone())
            nxt = torch.argmax(logits, dim=-1)
            nxt = nxt * unfinished + s.pad_token_id * (1 - unfinished)
            seq = torch.cat([seq, nxt[:, None]], dim=1)
            unfinished = unfinished & (nxt != s.eos_token_id).long()
            cur = nxt[:, None]
            if unfinished.max() == 0 or seq.shape[1] >= max_length:
                break
        return seq.numpy()

    # ------------------------------------------------------------------------------------------
    def beam_search(self, input_ids, bbox, pixel_values, attention_mask=None, num_beams=5, max_length=512,
                    length_penalty=1.0, early_stopping=False, e1=None):
        """Beam search over this oracle's KV-cached decoder (see `beam_search_core`)."""
        s = self.s
        input_ids = _t(input_ids).long()
        if attention_mask is None:
            attention_mask = self.default_generation_mask(input_ids)
        enc, mask = self.encode(input_ids, bbox, pixel_values, attention_mask)
        enc, mask = self.fuse_e1(enc, mask, e1)
        B = input_ids.shape[0]
        K = num_beams
        enc = enc.repeat_interleave(K, dim=0)
        mask = mask.repeat_interleave(K, dim=0)
        xkv = self.cross_kv(enc)
        st = {"kv": None}

        def logits_fn(running_seq, cur_len):
            cur = running_seq[:, :, cur_len - 1].reshape(B * K, 1)
            hid, st["kv"] = self.decoder_stack(cur, mask, xkv, st["kv"], cur_len - 1)
            return self.lm_logits(hid[:, -1:, :])[:, 0, :]

        def reorder_fn(bidx):
            # cache_utils.py:100-104 DynamicLayer.reorder_cache = index_select(0, beam_idx)
            st["kv"] = [(k.index_select(0, bidx), v.index_select(0, bidx)) for k, v in st["kv"]]

        return beam_search_core(logits_fn, reorder_fn, B, K, s.vocab_size, max_length, s.pad_token_id,
                                s.eos_token_id, s.decoder_start_token_id, length_penalty, early_stopping)


def beam_search_core(logits_fn, reorder_fn, B, K, V, max_length, pad_id, eos_id, start_id,
                     length_penalty=1.0, early_stopping=False):
    """gen:3208-3525 `_beam_search` (vectorised 5.15 form) restated: beams_to_keep = 2·num_beams (gen:3286),
    log_softmax + running scores and top-k over K·V (gen:3388,3418-3420,3077-3130), running beams for the
    next iteration (gen:3131-3152), finished-beam merge (gen:3153-3206), cache reorder by the selected
    beams (gen:3479-3485), early-stop heuristic (gen:3008-3053) and loop condition (gen:3055-3075).
    `logits_fn(running_seq [B,K,max_length], cur_len) -> [B*K, V]`; `reorder_fn(beam_idx [B*K])`.
    Returns (ids [B, T'], best finished score [B])."""
    keep = 2 * K
    top_mask = torch.cat([torch.ones(K, dtype=torch.bool), torch.zeros(keep - K, dtype=torch.bool)])
    # gen:3319 `output_fill_value = pad_token_id or eos_token_id[0]`: pad id 0 is falsy, so stock 5.15 fills
    # unfinished tails with EOS, not pad (a quirk the HIP path reproduces).
    fill_id = pad_id if pad_id else eos_id
    running_seq = torch.full((B, K, max_length), fill_id, dtype=torch.long)
    running_seq[:, :, 0] = start_id
    sequences = running_seq.clone()
    running_scores = torch.zeros(B, K, dtype=F32)
    running_scores[:, 1:] = -1e9
    beam_scores = torch.full((B, K), -1e9, dtype=F32)
    is_fin = torch.zeros(B, K, dtype=torch.bool)
    heur = torch.ones(B, 1, dtype=torch.bool)
    run_idx = torch.full((B, K, max_length - 1), -1, dtype=torch.int32)
    beam_idx_out = run_idx.clone()
    cur_len = 1

    def gather(t, idx):
        while idx.dim() < t.dim():
            idx = idx.unsqueeze(-1)
        return torch.take_along_dim(t, idx, dim=1)

    while True:
        logits = logits_fn(running_seq, cur_len).to(F32)
        logp = torch.log_softmax(logits, dim=-1).view(B, K, V) + running_scores[:, :, None]
        logp = logp.reshape(B, K * V)
        topv, topi = torch.topk(logp, k=keep)
        src_beam = topi // V
        top_run_idx = gather(run_idx, src_beam)
        top_seq = gather(running_seq, src_beam)
        top_seq[:, :, cur_len] = topi % V
        top_run_idx[:, :, cur_len - 1] = (src_beam + torch.arange(B).view(-1, 1) * K).to(torch.int32)
        hits = (top_seq[:, :, cur_len] == eos_id) | (cur_len + 1 >= max_length)
        # e. running beams for the next iteration
        run_lp = topv + hits.to(F32) * -1.0e9
        nxt = torch.topk(run_lp, k=K)[1]
        running_seq = gather(top_seq, nxt)
        running_scores = gather(run_lp, nxt)
        run_idx = gather(top_run_idx, nxt)
        # f. finished beams
        just = hits & top_mask[None, :]
        fin_lp = topv / (cur_len ** length_penalty)          # (cur_len + 1 - decoder_prompt_len), prompt = 1
        fin_lp = fin_lp + (torch.all(is_fin, dim=-1, keepdim=True) & (early_stopping is True)).to(F32) * -1.0e9
        fin_lp = fin_lp + (~heur).to(F32) * -1.0e9
        fin_lp = fin_lp + (~just) * -1.0e9
        m_seq = torch.cat([sequences, top_seq], dim=1)
        m_sc = torch.cat([beam_scores, fin_lp], dim=1)
        m_bi = torch.cat([beam_idx_out, top_run_idx], dim=1)
        m_fin = torch.cat([is_fin, just], dim=1)
        sel = torch.topk(m_sc, k=K)[1]
        sequences = gather(m_seq, sel)
        beam_scores = gather(m_sc, sel)
        beam_idx_out = gather(m_bi, sel)
        is_fin = gather(m_fin, sel)
        # g. cache reorder + loop condition
        reorder_fn(run_idx[:, :, cur_len - 1].reshape(-1).long())
        cur_len += 1
        best_possible = running_scores[:, :1] / ((cur_len - 1) ** length_penalty)
        worst_fin = torch.where(is_fin, torch.min(beam_scores, dim=1, keepdim=True)[0], torch.tensor(-1.0e9))
        heur = heur & torch.any(best_possible > worst_fin, dim=-1, keepdim=True)
        cont = torch.any(heur) & ~(torch.all(is_fin) & (early_stopping is True)) & ~torch.all(hits)
        if not bool(cont):
            break
    seqs = sequences[:, 0, :]
    bi = beam_idx_out[:, 0, :]
    gen_len = int(((bi + 1).bool()).sum(dim=1).max())
    return seqs[:, :1 + gen_len].numpy(), beam_scores[:, 0].numpy()
