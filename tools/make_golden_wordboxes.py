"""Mints tests/golden/host_wordboxes.json: inputs and outputs of the REFERENCE's own word-box / cell-text helpers
(/root/reference/markushgrapher/core/common/data_preprocessing.py: split_bounding_box_for_words, prepare_cells_to_text,
with check_max_values / normalize_bbox_format from core/common/utils.py), executed unmodified in the build container.

The two files are loaded by path under their own module names.  What they import and this image lacks is stubbed with EMPTY
modules (`torchvision`, `matplotlib`, and `torch._utils._accumulate`, removed in torch 2.10): none of the four
functions touches those names.  The tokenizer is a deterministic stand-in with the one method the functions call
(`tokenize(str) -> pieces`, sentencepiece-style "▁" word prefix) - the reference's sentencepiece model is not available
offline; the same stand-in drives the build's port in tests/test_assembly.py, so the box arithmetic, the filters (>500 px drop,
whitespace pieces) and the token-budget control flow are what is pinned.  Only data (inputs / outputs) is written.
    python tools/make_golden_wordboxes.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/markushgrapher/core/common"


class PieceTokenizer:
    """tokenize(): whitespace words -> '▁' + word, words longer than 5 characters split into pieces of 4; a run of two or more
    spaces yields a bare '▁' piece (which the reference measures as one character and then skips as whitespace-only)."""

    def tokenize(self, text):
        out = []
        for i, w in enumerate(text.split(" ")):
            if w == "":
                if i > 0:
                    out.append("▁")
                continue
            parts = [w[j:j + 4] for j in range(0, len(w), 4)] if len(w) > 5 else [w]
            out.append("▁" + parts[0])
            out.extend(parts[1:])
        return out


def load_ref():
    import torch
    for name in ("matplotlib", "matplotlib.pyplot", "torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    if not hasattr(torch._utils, "_accumulate"):
        torch._utils._accumulate = lambda *a, **k: None
    for pkg in ("markushgrapher", "markushgrapher.core", "markushgrapher.core.common"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules.setdefault(pkg, m)
    mods = {}
    for short in ("utils", "data_preprocessing"):
        name = f"markushgrapher.core.common.{short}"
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, short + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        mods[short] = mod
    return mods["data_preprocessing"], mods["utils"]


def main():
    dp, ut = load_ref()
    tok = PieceTokenizer()
    rng = np.random.default_rng(20260929)
    vocab = ["C", "N", "R1", "R2", "methyl", "ethyl", "phenyl", "=", "H", "Cl", "cycloalkyl", "X", "or", "is", "selected", "from",
             "1-3", "(CH2)n", "O", "alkoxy"]
    split_cases = []
    for _ in range(12):
        n = int(rng.integers(1, 6))
        sent = " ".join(rng.choice(vocab, n))
        if rng.random() < 0.3:
            sent = sent.replace(" ", "  ", 1)
        box = [float(x) for x in np.round(rng.random(2) * 300, 3)]
        box = box + [box[0] + float(np.round(rng.random() * 200 + 1, 3)), box[1] + float(np.round(rng.random() * 30 + 1, 3))]
        words, boxes = dp.split_bounding_box_for_words(sent, box, tok)
        split_cases.append({"sentence": sent, "bbox": box, "words": words, "boxes": [list(b) for b in boxes]})
    cell_cases = []
    for ci in range(8):
        ncell = int(rng.integers(1, 9)) if ci < 6 else 90
        cells = []
        for _ in range(ncell):
            n = int(rng.integers(1, 5))
            text = " ".join(rng.choice(vocab, n)) if rng.random() > 0.1 else "   "
            x0, y0 = rng.random(2) * 0.95
            x1, y1 = min(1.0, x0 + rng.random() * 0.2 + 0.01), min(1.0, y0 + rng.random() * 0.05 + 0.01)
            if rng.random() < 0.15:
                x1 = min(1.0, x0 + 0.9)           # reaches beyond 500 px of a 512 px page: dropped by check_max_values
            cells.append({"text": text, "bbox": [float(x0), float(y0), float(x1), float(y1)]})
        for norm in (True, False):
            kw = dict(w=512, h=512, normalize_bbox=norm, max_sequence_length=512 if ci < 7 else 64)
            words, boxes, tidx = dp.prepare_cells_to_text(cells, tok, **kw)
            cell_cases.append({"cells": cells, "kwargs": kw, "words": words, "boxes": [list(b) for b in boxes], "token_idx": tidx})
    misc = {"check_max_values": [[b, bool(ut.check_max_values(b))] for b in ([1, 2, 3, 4], [0, 0, 500, 500], [0, 0, 500.5, 3], [501, 0, 0, 0])],
            "normalize_bbox_format": [[b, list(ut.normalize_bbox_format(b, 512, 512))] for b in ([0, 0, 512, 512], [10.7, 20.2, 300.9, 511.9])],
            "estimate_word_width": [[w, dp.estimate_word_width(w)] for w in ("▁", "▁C", "ethyl", "▁cyclo")],
            "normalText": [[t, dp.normalText(t)] for t in (" a ", 3.0, 2.5, "▁x")]}
    out = {"source": "reference data_preprocessing.py / utils.py executed unmodified with stub modules (tools/make_golden_wordboxes.py)",
           "tokenizer": "PieceTokenizer of this script", "split_bounding_box_for_words": split_cases, "prepare_cells_to_text": cell_cases,
           "misc": misc}
    with open(os.path.join(ROOT, "tests", "golden", "host_wordboxes.json"), "w") as f:
        json.dump(out, f)
    print("wrote", len(split_cases), "split cases,", len(cell_cases), "cell cases")


if __name__ == "__main__":
    main()
