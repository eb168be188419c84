"""Mints tests/golden/host_collator.json: inputs and outputs of the REFERENCE's own DataCollator
(/root/reference/markushgrapher/core/trainers/data_collator.py) on seeded ragged features.  The file is loaded by path
(its package __init__ pulls in libraries that do not exist in this image; the collator file itself needs only torch
and transformers), executed unmodified, and only data (inputs / outputs) is written.  Run in the build container:
    python tools/make_golden_host.py
"""
import importlib.util
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/markushgrapher/core/trainers/data_collator.py"


def load_ref():
    spec = importlib.util.spec_from_file_location("_ref_data_collator", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_features(rng, n, lens, dec_lens, with_char, with_image_key):
    feats = []
    for i in range(n):
        L, T = lens[i], dec_lens[i]
        f = {
            "input_ids": torch.tensor(rng.integers(3, 32000, L), dtype=torch.long),
            "attention_mask": torch.ones(L, dtype=torch.long),
            "bbox": torch.tensor(rng.random((L, 4)), dtype=torch.float32),
            "labels": torch.tensor(rng.integers(3, 32000, T), dtype=torch.long),
            "decoder_attention_mask": torch.ones(T, dtype=torch.long),
            "visual_seg_data": torch.tensor(rng.random((4, 4)), dtype=torch.float32),
        }
        if with_char:
            C = int(rng.integers(5, 40))
            f["char_ids"] = torch.tensor(rng.integers(0, 200, C), dtype=torch.long)
            f["char_seg_data"] = torch.tensor(rng.integers(0, 9, C), dtype=torch.long)
        if with_image_key:
            f["image"] = None
        f["pixel_values"] = torch.tensor(rng.random((3, 4, 4)), dtype=torch.float32)
        feats.append(f)
    return feats


def to_json(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = {"dtype": str(v.dtype).replace("torch.", ""), "shape": list(v.shape), "data": v.flatten().tolist()}
        else:
            out[k] = None
    return out


def main():
    ref = load_ref()
    rng = np.random.default_rng(20260928)
    cases = []
    specs = [
        dict(n=3, lens=[5, 9, 2], dec=[4, 7, 3], kw=dict(max_length=12, max_length_decoder=6, max_length_char=20), char=False, img=False),
        dict(n=4, lens=[1, 16, 8, 8], dec=[2, 2, 9, 1], kw=dict(max_length=8, max_length_decoder=8, max_length_char=16), char=True, img=True),
        dict(n=1, lens=[30], dec=[12], kw=dict(max_length=24, max_length_decoder=10, max_length_char=1536), char=True, img=False),
        dict(n=2, lens=[6, 6], dec=[5, 5], kw=dict(max_length=6, max_length_decoder=5, max_length_char=8), char=False, img=True),
    ]
    for sp in specs:
        feats = make_features(rng, sp["n"], sp["lens"], sp["dec"], sp["char"], sp["img"])
        inputs = [to_json(f) for f in feats]
        out = ref.DataCollator(**sp["kw"])(feats)
        cases.append({"kwargs": sp["kw"], "features": inputs, "key_order": list(feats[0].keys()), "batch": to_json(out)})
    # list inputs (pad_sequence_native's non-tensor branch) and the None-feature placeholder
    pads = []
    for seq, tgt, pad in ([[1, 2, 3], 6, 0], [[1, 2, 3, 4, 5], 3, -100], [[[1, 2, 3, 4]], 3, [0, 0, 0, 0]], [[], 2, 7]):
        pads.append({"seq": seq, "target_len": tgt, "pad_value": pad, "out": to_json({"o": ref.pad_sequence_native(seq, tgt, pad)})["o"]})
    placeholder = to_json(ref.DataCollator()([None]))
    with open(os.path.join(ROOT, "tests", "golden", "host_collator.json"), "w") as f:
        json.dump({"source": "reference DataCollator executed unmodified (tools/make_golden_host.py)", "cases": cases,
                   "pad_sequence_native": pads, "placeholder": placeholder}, f)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
