"""Shapes, HF state-dict layout and deterministic recipe weights of the ChemicalOCR stage.

The reference loads its OCR checkpoint with `AutoModelForVision2Seq.from_pretrained(model_path)` (an Idefics3 / SmolDocling-class
vision-language model: SigLIP-style vision tower, pixel-shuffle connector, Llama-style text model) and calls
`model.generate(**inputs, max_new_tokens=4096, do_sample=False)` on one 512-px page at a time
(ref: markushgrapher/ocr/chemical_ocr.py:76-84, 366-392).  The checkpoint itself (`checkpoints/chemicalocr_v3`) is not in the
reference tree: the "smoldocling" preset below is the published SmolDocling-256M geometry and is INFERRED, like SURVEY.md says;
key names are those of stock `Idefics3ForConditionalGeneration` (transformers 5.15, models/idefics3/modeling_idefics3.py).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, List, Tuple

import numpy as np

from .synth import uniform_pm1, round_bf16


@dataclass(frozen=True)
class OcrShape:
    # vision tower (Idefics3VisionConfig)
    v_hidden: int = 768
    v_inter: int = 3072
    v_layers: int = 12
    v_heads: int = 12
    image_size: int = 512
    patch_size: int = 16
    v_eps: float = 1e-6
    # text model (LlamaConfig)
    t_hidden: int = 576
    t_inter: int = 1536
    t_layers: int = 30
    t_heads: int = 9
    t_kv_heads: int = 3
    vocab: int = 49280
    rms_eps: float = 1e-5
    rope_theta: float = 100000.0
    # glue (Idefics3Config)
    scale_factor: int = 4
    image_token_id: int = 49190
    eos_token_id: int = 49279
    pad_token_id: int = 2
    tie_word_embeddings: bool = False
    eos_extra: Tuple[int, ...] = ()      # further stop tokens (generation_config.json may list several EOS ids); at most 3

    @property
    def patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2

    @property
    def image_seq_len(self) -> int:
        return self.patches // (self.scale_factor ** 2)

    def as_dict(self):
        return asdict(self)


PRESETS = {
    # SmolDocling-256M geometry (INFERRED for the reference's ChemicalOCR checkpoint)
    "smoldocling": OcrShape(),
    # parity fixture: every code path of the big one at a size the CPU oracle and the SIMT emulator finish in seconds
    "tiny": OcrShape(v_hidden=64, v_inter=128, v_layers=2, v_heads=1, image_size=64, patch_size=16, t_hidden=128, t_inter=256,
                     t_layers=2, t_heads=2, t_kv_heads=1, vocab=320, scale_factor=2, image_token_id=300, eos_token_id=1,
                     pad_token_id=2),
}


def state_dict_spec(s: OcrShape) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(HF key, shape, kind) in stock order; kind in {linear, bias, norm_w, norm_b, embed, conv}."""
    out = []
    v = "model.vision_model."
    out.append((v + "embeddings.patch_embedding.weight", (s.v_hidden, 3, s.patch_size, s.patch_size), "conv"))
    out.append((v + "embeddings.patch_embedding.bias", (s.v_hidden,), "bias"))
    out.append((v + "embeddings.position_embedding.weight", (s.patches, s.v_hidden), "embed"))
    for i in range(s.v_layers):
        p = f"{v}encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            out.append((p + f"self_attn.{n}.weight", (s.v_hidden, s.v_hidden), "linear"))
            out.append((p + f"self_attn.{n}.bias", (s.v_hidden,), "bias"))
        out.append((p + "layer_norm1.weight", (s.v_hidden,), "norm_w"))
        out.append((p + "layer_norm1.bias", (s.v_hidden,), "norm_b"))
        out.append((p + "mlp.fc1.weight", (s.v_inter, s.v_hidden), "linear"))
        out.append((p + "mlp.fc1.bias", (s.v_inter,), "bias"))
        out.append((p + "mlp.fc2.weight", (s.v_hidden, s.v_inter), "linear"))
        out.append((p + "mlp.fc2.bias", (s.v_hidden,), "bias"))
        out.append((p + "layer_norm2.weight", (s.v_hidden,), "norm_w"))
        out.append((p + "layer_norm2.bias", (s.v_hidden,), "norm_b"))
    out.append((v + "post_layernorm.weight", (s.v_hidden,), "norm_w"))
    out.append((v + "post_layernorm.bias", (s.v_hidden,), "norm_b"))
    out.append(("model.connector.modality_projection.proj.weight", (s.t_hidden, s.v_hidden * s.scale_factor ** 2), "linear"))
    t = "model.text_model."
    out.append((t + "embed_tokens.weight", (s.vocab, s.t_hidden), "embed"))
    kvd = s.t_kv_heads * 64
    for i in range(s.t_layers):
        p = f"{t}layers.{i}."
        out.append((p + "self_attn.q_proj.weight", (s.t_hidden, s.t_hidden), "linear"))
        out.append((p + "self_attn.k_proj.weight", (kvd, s.t_hidden), "linear"))
        out.append((p + "self_attn.v_proj.weight", (kvd, s.t_hidden), "linear"))
        out.append((p + "self_attn.o_proj.weight", (s.t_hidden, s.t_hidden), "linear"))
        out.append((p + "mlp.gate_proj.weight", (s.t_inter, s.t_hidden), "linear"))
        out.append((p + "mlp.up_proj.weight", (s.t_inter, s.t_hidden), "linear"))
        out.append((p + "mlp.down_proj.weight", (s.t_hidden, s.t_inter), "linear"))
        out.append((p + "input_layernorm.weight", (s.t_hidden,), "norm_w"))
        out.append((p + "post_attention_layernorm.weight", (s.t_hidden,), "norm_w"))
    out.append((t + "norm.weight", (s.t_hidden,), "norm_w"))
    if not s.tie_word_embeddings:
        out.append(("lm_head.weight", (s.vocab, s.t_hidden), "embed"))
    return out


def recipe_state_dict(s: OcrShape, seed: int = 20260929, gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Counter-based weights (bf16-exact fp32): linear ~ U(-1,1) * gain * sqrt(3 / fan_in), embeddings U(-1,1) * 0.5, norm weights
    1 + 0.1 U, biases 0.1 U.  Regenerated identically on the GPU box; nothing is stored."""
    sd = {}
    for name, shape, kind in state_dict_spec(s):
        u = uniform_pm1("ocr/" + name, shape, seed)
        if kind in ("linear", "conv"):
            fan_in = int(np.prod(shape[1:]))
            w = u * np.float32(gain * np.sqrt(3.0 / fan_in))
        elif kind == "embed":
            w = u * np.float32(0.5)
        elif kind == "norm_w":
            w = np.float32(1.0) + np.float32(0.1) * u
        else:
            w = np.float32(0.1) * u
        sd[name] = round_bf16(w.astype(np.float32))
    return sd


def synth_inputs(s: OcrShape, B: int, prompt_text_tokens: int = 12, seed: int = 20260929, n_img: int = 1):
    """The reference's input contract for one page per sample (chemical_ocr.py:366-373 with the Idefics3 processor): prompt ids
    = [text..., <fake>, <image> x image_seq_len, <fake>, text...] and pixel_values [B][1][3][I][I] in [-1, 1] (full image, no
    padding: pixel_attention_mask all ones).  Token ids avoid the image / eos / pad ids."""
    from .synth import randint
    n_frames = n_img
    n_img = s.image_seq_len * n_frames            # <image> tokens per sequence (frames back to back inside one wrapper pair)
    head = prompt_text_tokens // 2
    tail = prompt_text_tokens - head
    hi = min(s.vocab, s.image_token_id) - 2
    ids = np.zeros((B, head + 1 + n_img + 1 + tail), np.int64)
    fake = hi + 1 if hi + 1 != s.image_token_id else hi
    for b in range(B):
        t = randint(f"ocr/prompt/{b}", prompt_text_tokens, 3, hi - 1, seed)
        ids[b, :head] = t[:head]
        ids[b, head] = fake
        ids[b, head + 1:head + 1 + n_img] = s.image_token_id
        ids[b, head + 1 + n_img] = fake
        ids[b, head + 2 + n_img:] = t[head:]
    pix = uniform_pm1("ocr/pixels", (B, n_frames, 3, s.image_size, s.image_size), seed).astype(np.float32)
    # page-like: mostly white with dark strokes (smooth-ish blocks), still fully deterministic
    blocks = uniform_pm1("ocr/blocks", (B, n_frames, 1, s.image_size // 8, s.image_size // 8), seed)
    pix = np.where(np.repeat(np.repeat(blocks, 8, -1), 8, -2) > 0.7, pix, np.float32(1.0) - np.float32(0.05) * np.abs(pix))
    return ids, pix.astype(np.float32)


def synth_pixel_mask(s: OcrShape, B: int, n_img: int = 1, seed: int = 20260929):
    """pixel_attention_mask [B][n][I][I] of non-square pages as the Idefics3 processor pads them: valid region = top-left
    (rows < h, columns < w) with h, w not multiples of the patch size for some frames; frame 0 of sequence 0 stays full."""
    from .synth import randint
    I = s.image_size
    m = np.zeros((B, n_img, I, I), bool)
    hs = randint("ocr/mask_h", B * n_img, I // 4, 3 * I // 4, seed).reshape(B, n_img)      # at least a quarter of the patch rows / columns
    ws = randint("ocr/mask_w", B * n_img, I // 4, 3 * I // 4, seed).reshape(B, n_img)      # of a padded frame are fully masked
    for b in range(B):
        for j in range(n_img):
            h, w = (I, I) if (b == 0 and j == 0) else (int(hs[b, j]), int(ws[b, j]))
            if (b + j) % 2:
                h = I                  # landscape / portrait alternate: one side stays full, as longest-edge resizing gives
            else:
                w = I if not (b == 0 and j == 0) else I
            m[b, j, :h, :w] = True
    return m


# ---------------------------------------------------------------------------------------------------------------------------
# Scripted OCR output (end-to-end tests and the configs[4] bench loop): no ChemicalOCR checkpoint and no tokenizer model exist
# offline, and a random-weight model emits noise that parses to no cells.  To let REAL cell strings flow from the OCR stage into
# the VTL stage, the lm_head / embedding of an otherwise ordinary recipe model are set so that greedy decoding walks a scripted
# chain of tokens per page (lm_head[succ(t)] = direction of embed[t]; the embeddings dominate the residual stream).  The compute
# is that of any model of the shape - only WHICH token wins is scripted.  The id -> piece table is the stand-in tokenizer.
# ---------------------------------------------------------------------------------------------------------------------------
def script_texts(s: OcrShape, texts: List[str], piece_len: int = 3):
    """-> (id_to_piece [vocab], chains: per text the token ids it is emitted as (EOS last), starts: the prompt-final token selecting
    each chain).  Every chain position has its own token id, so a chain never revisits a token."""
    reserved = {s.eos_token_id, s.pad_token_id, s.image_token_id}
    free = [i for i in range(3, s.vocab) if i not in reserved]
    need = sum(1 + (len(t) + piece_len - 1) // piece_len for t in texts)
    if need > len(free):
        raise ValueError(f"{need} scripted tokens do not fit the vocabulary ({len(free)} free ids)")
    id_to_piece = [f"<t{i}>" for i in range(s.vocab)]
    chains, starts, nxt = [], [], 0
    for k, text in enumerate(texts):
        start = free[nxt]; nxt += 1
        id_to_piece[start] = f"<start_{k}>"
        ids = []
        for i in range(0, len(text), piece_len):
            id_to_piece[free[nxt]] = text[i:i + piece_len]
            ids.append(free[nxt]); nxt += 1
        chains.append(ids + [s.eos_token_id])
        starts.append(start)
    return id_to_piece, chains, starts


def scripted_state_dict(s: OcrShape, chains, starts, gain: float = 0.5, embed_scale: float = 48.0, seed: int = 20260929):
    sd = recipe_state_dict(s, seed=seed, gain=gain)
    emb = round_bf16(sd["model.text_model.embed_tokens.weight"] * np.float32(embed_scale))
    sd["model.text_model.embed_tokens.weight"] = emb
    head = round_bf16(uniform_pm1("ocr/script/lm_head", emb.shape, seed) * np.float32(0.05))
    eos_row = np.zeros(emb.shape[1], np.float32)
    for start, chain in zip(starts, chains):
        prev = start
        for tid in chain[:-1]:
            head[tid] = round_bf16(emb[prev] / np.float32(embed_scale))
            prev = tid
        eos_row += emb[prev] / np.float32(embed_scale)      # all chains end in the one EOS token: its row points at every last piece
    head[s.eos_token_id] = round_bf16(eos_row / np.float32(max(1.0, np.sqrt(len(chains)) / 2)))
    sd["lm_head.weight"] = head
    return sd


def scripted_prompts(s: OcrShape, chains, starts, prompt_text_tokens: int = 10):
    """input_ids [pages][L]: the stock prompt layout with the page's start token last; no prompt token is part of a chain."""
    ids, _ = synth_inputs(s, len(starts), prompt_text_tokens=prompt_text_tokens)
    used = np.array(sorted({t for c in chains for t in c} | set(starts)))
    neutral = next(i for i in range(s.vocab - 1, 2, -1) if i not in set(used.tolist()) and i not in (s.image_token_id, s.eos_token_id, s.pad_token_id))
    text_pos = ids != s.image_token_id
    ids = np.where(np.isin(ids, used) & text_pos, neutral, ids)
    for b, st in enumerate(starts):
        ids[b, -1] = st
    return ids


def detokenize(id_to_piece, row, eos_id, pad_id):
    """Stand-in for processor.batch_decode(generated_ids[:, prompt_len:], skip_special_tokens=True) (ref: chemical_ocr.py:386-390)."""
    out = []
    for t in row:
        t = int(t)
        if t == eos_id:
            break
        if t != pad_id:
            out.append(id_to_piece[t])
    return "".join(out)


_CELL_WORDS = ["R1", "R2", "R3", "R4", "alkyl", "group", "hydrogen", "atom", "halogen", "represents", "wherein", "and", "or", "same", "different",
               "each", "may", "be", "methyl", "ethyl", "phenyl", "alkoxy", "C1-C6", "OH", "NH", "Cl", "Br", "Me", "Et", "Ph", "Ar", "Het", "ring",
               "aryl", "selected", "from", "of", "a", "is", "="]


def synth_cell_text(n_cells: int, seed: int, name: str = "page") -> str:
    """A page's OCR output in the current grammar (`<ocr>x1>y1>x2>y2>text\n...</ocr>`, ref: chemical_ocr.py:165-199): n_cells cells of 1-6
    words in boxes on the 500-unit grid, IP5-M-like (10-120 cells per page)."""
    from .synth import randint
    r = randint(f"ocr/cells/{name}", n_cells * 12, 0, 1 << 20, seed).reshape(n_cells, 12)
    lines = []
    for c in r:
        x0, y0 = int(c[0] % 420), int(c[1] % 470)
        w, h = 20 + int(c[2] % 60), 10 + int(c[3] % 15)
        nw = 1 + int(c[4] % 6)
        words = " ".join(_CELL_WORDS[int(c[5 + j] % len(_CELL_WORDS))] for j in range(nw))
        lines.append(f"{x0}>{y0}>{min(x0 + w, 499)}>{min(y0 + h, 499)}>{words}")
    return "<ocr>" + "\n".join(lines) + "</ocr>"
