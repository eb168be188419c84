"""Mint the G3 "trained-tiny" fixture weights: a tiny stock-UDOP model trained for a few hundred
CPU steps on a synthetic class->sequence task so greedy / beam token ids have top-1/top-2 margins far above bf16 noise and
rows emit EOS at different steps.  Runs ONLY in the build container (imports stock transformers); writes
tests/golden/g3_weights.npz (bf16-exact fp32 weights).  The stock model never travels.

    python tools/train_tiny.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from markushgrapher_amd import synth  # noqa: E402
from tools.stock import stock_model  # noqa: E402


NUM_CLASSES = 16


def class_sequences(shape):
    """The memorised class -> output sequence table (lengths 2..9, shared prefixes between some classes so beam
    search has close competitors that finish at different steps)."""
    seqs = []
    for c in range(NUM_CLASSES):
        n = 2 + (c * 5) % 8
        toks = synth.randint(f"cls.seq{c // 2}", 9, 20, shape.vocab_size - 1, 7)[:n].copy()
        if c % 2 == 1 and n > 2:
            toks[-1] = synth.randint(f"cls.alt{c}", 1, 20, shape.vocab_size - 1, 7)[0]
        seqs.append(toks)
    return seqs


def copy_task_batch(shape, B, seed, n_min=2, n_max=9, T=12):
    """text = [2 question tokens, box 0][sep, box 1][n copies of a class token, random word boxes][sep, box 1];
    labels = that class's memorised sequence + EOS, padded with -100
    (padding rules ref: core/trainers/data_collator.py:55-61,106-108)."""
    seqs = class_sequences(shape)
    n = synth.randint("copy.n", B, n_min, n_max, seed)
    cls = synth.randint("copy.c", B, 0, NUM_CLASSES - 1, seed)
    L = int(n.max()) + 4
    ids = np.zeros((B, L), np.int64)
    bbox = np.zeros((B, L, 4), np.float32)
    mask = np.zeros((B, L), np.int64)
    labels = np.full((B, T), -100, np.int64)
    for b in range(B):
        k = int(n[b])
        ids[b, 0:2] = [7, 8]
        ids[b, 2] = shape.eos_token_id
        ids[b, 3:3 + k] = 3 + int(cls[b])
        ids[b, 3 + k] = shape.eos_token_id
        mask[b, :4 + k] = 1
        u = synth.uniform01(f"copy.b{b}", k * 4, seed).reshape(k, 4)
        x0, y0 = u[:, 0] * 0.85, u[:, 1] * 0.85
        bb = np.stack([x0, y0, x0 + 0.02 + u[:, 2] * 0.1, y0 + 0.02 + u[:, 3] * 0.1], -1).astype(np.float32)
        bbox[b, 2] = 1.0
        bbox[b, 3:3 + k] = np.clip(bb, 0, 1)
        bbox[b, 3 + k] = 1.0
        sq = seqs[int(cls[b])]
        labels[b, :len(sq)] = sq
        labels[b, len(sq)] = shape.eos_token_id
    pages = synth.synth_pages_u8(B, shape.image_size, seed)
    pv = synth.pages_to_pixel_values(pages, shape.image_size)
    return {"input_ids": ids, "bbox": bbox, "attention_mask": mask, "pixel_values": pv, "labels": labels}


def main(steps=500, out="tests/golden/g3_weights.npz"):
    torch.manual_seed(0)
    shape = synth.SHAPES["tiny"]
    sd0 = synth.recipe_state_dict(shape, gain=1.0)
    m = stock_model(shape, sd0)
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=3e-3, weight_decay=0.0)
    for step in range(steps):
        b = copy_task_batch(shape, 32, seed=1000 + step)
        t = {k: torch.from_numpy(v) for k, v in b.items()}
        dam = (t["labels"] != -100).long()
        out_ = m(input_ids=t["input_ids"], bbox=t["bbox"], pixel_values=t["pixel_values"],
                 attention_mask=t["attention_mask"], labels=t["labels"], decoder_attention_mask=dam)
        opt.zero_grad()
        out_.loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        for g in opt.param_groups:
            g["lr"] = 3e-3 * min(1.0, (step + 1) / 50) * (0.5 ** (step / 250))
        opt.step()
        if step % 50 == 0 or step == steps - 1:
            print(step, float(out_.loss), flush=True)
    m.eval()
    full = m.state_dict()
    sd = {}
    for key, shp, _ in synth.state_dict_spec(shape):
        src = key
        if key == "encoder.relative_bias.biases.0.relative_attention_bias.weight":
            src = "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"
        sd[key] = synth.round_bf16(full[src].detach().numpy().astype(np.float32))
        assert sd[key].shape == tuple(shp), (key, sd[key].shape, shp)
    np.savez_compressed(out, **sd)
    print("wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    main()
