"""Host-side rows f-3 (batching) and f-4 (ids -> text): `markushgrapher_amd/assembly.py`.

`DataCollator` is checked against outputs of the reference's own class (tests/golden/host_collator.json, minted by
tools/make_golden_host.py).  The word-box / cell-text / decode helpers are known-answer tests written from the
reference's documented behaviour (their modules cannot be imported in the build image: parity unpinned)."""
import json
import os

import numpy as np
import pytest
import torch

from markushgrapher_amd import assembly as A
from tests.conftest import GOLDEN

DT = {"int64": torch.int64, "int32": torch.int32, "float32": torch.float32}


def _t(j):
    if j is None:
        return None
    return torch.tensor(j["data"], dtype=DT[j["dtype"]]).reshape(j["shape"])


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLDEN, "host_collator.json")) as f:
        return json.load(f)


def test_data_collator_matches_reference(gold):
    for case in gold["cases"]:
        feats = [{k: _t(f[k]) for k in case["key_order"]} for f in case["features"]]
        got = A.DataCollator(**case["kwargs"])(feats)
        want = {k: _t(v) for k, v in case["batch"].items()}
        assert list(got.keys()) == list(want.keys())
        for k, w in want.items():
            g = got[k]
            if w is None:
                assert g is None
                continue
            assert g.dtype == w.dtype and g.shape == w.shape, k
            assert torch.equal(g, w), k


def test_pad_sequence_native_matches_reference(gold):
    for c in gold["pad_sequence_native"]:
        got = A.pad_sequence_native(c["seq"], c["target_len"], c["pad_value"])
        want = _t(c["out"])
        assert got.dtype == want.dtype and list(got.shape) == list(want.shape) and torch.equal(got, want)
    ph = A.DataCollator()([None])
    assert torch.equal(ph["placeholder"], _t(gold["placeholder"]["placeholder"]))


def test_collate_for_generate_pads_to_longest():
    feats = [{"input_ids": torch.arange(3, 3 + n), "bbox": torch.rand(n, 4), "pixel_values": torch.zeros(3, 2, 2)} for n in (4, 7, 1)]
    b = A.collate_for_generate(feats)
    assert b["input_ids"].shape == (3, 7) and b["bbox"].shape == (3, 7, 4) and b["pixel_values"].shape == (3, 3, 2, 2)
    assert b["attention_mask"].sum(1).tolist() == [4, 7, 1]
    assert b["input_ids"][2].tolist() == [3, 0, 0, 0, 0, 0, 0] and float(b["bbox"][2, 1:].abs().sum()) == 0.0


class _Tok:
    """whitespace 'sentencepiece': every word becomes one piece with the word-start marker; words longer than 4
    characters are cut into 4-character pieces"""

    def tokenize(self, s):
        out = []
        for w in s.split():
            w = A.SP + w
            out.extend(w[i:i + 5] if i == 0 else w[i:i + 4] for i in ([0] + list(range(5, len(w), 4))))
        return out


def test_word_boxes_are_proportional_and_contiguous():
    assert A.estimate_word_width(A.SP) == 12 and A.estimate_word_width(A.SP + "ab") == 24 and A.estimate_word_width("abc") == 36
    pieces, boxes = A.split_bounding_box_for_words("R1 represents", (10.0, 5.0, 130.0, 25.0), _Tok())
    assert pieces == [A.SP + "R1", A.SP + "repr", "esen", "ts"]
    # widths 2:4:4:2 of 120 px
    np.testing.assert_allclose([b[2] - b[0] for b in boxes], [20.0, 40.0, 40.0, 20.0])
    assert boxes[0][0] == 10.0 and abs(boxes[-1][2] - 130.0) < 1e-9
    assert all(boxes[i][2] == boxes[i + 1][0] for i in range(3)) and all(b[1] == 5.0 and b[3] == 25.0 for b in boxes)


def test_prepare_cells_skips_and_limits():
    cells = [{"text": "  ", "bbox": [0, 0, 1, 1]}, {"text": "R1 O", "bbox": [0.1, 0.1, 0.3, 0.2]},
             {"text": "far", "bbox": [0.5, 0.5, 1.2, 0.6]}]
    words, boxes, n = A.prepare_cells_to_text(cells, _Tok(), 400, 400, normalize_bbox=False)
    # third cell maps beyond 500 after the 0-500 normalisation -> dropped
    assert words == [A.SP + "R1", A.SP + "O"] and n == 2
    assert boxes[0] == (50, 50, 116, 100) and boxes[1][2] == 150
    words, boxes, n = A.prepare_cells_to_text(cells[1:2] * 100, _Tok(), 400, 400, True, max_sequence_length=20)
    # 15 short of the limit only the CURRENT cell is abandoned: every later cell still contributes its first piece
    # until the limit itself is reached (the reference's two nested breaks)
    assert n == 20 and len(words) == 20 and words[:6] == [A.SP + "R1", A.SP + "O"] * 2 + [A.SP + "R1"] * 2
    img = type("I", (), {"size": (200, 100)})()
    _, instr, w2, b2, labels = A.collate_item({"image": img, "cells": cells[1:2], "entities": {"question": "q?", "answer": 3.0}}, _Tok(), True)
    assert instr == "Question Answering. q?" and labels == ["3", "</s>"]
    np.testing.assert_allclose(b2[0], [0.1, 0.1, 0.1 + 0.2 * 2 / 3, 0.2])


def test_id_decoder_rules():
    vocabulary = {"<cxsmi>": "<other_0>", "</cxsmi>": "<other_1>", "C": "<other_2>", "<i>": "<other_3>", "</i>": "<other_4>"}
    inverse = {v: k for k, v in vocabulary.items()}
    toks = ["<pad>", "</s>", A.SP + "alkyl", "group", A.SP, "<loc_12>", "<other_0>", "<other_1>", "<other_2>", "<other_3>",
            "<other_4>", "<other_99>", "7"]
    d = A.IdDecoder(toks, vocabulary, inverse, encode_index=False)
    # other tokens carry their own trailing space; an ordinary piece gets one when the NEXT token has a marker / is "other"
    assert d.decode([6, 8, 8, 7]) == "<cxsmi> C C </cxsmi> "
    assert d.decode([2, 3, 2, 3]) == "alkylgroup alkylgroup"      # a space only before a piece that carries the marker
    assert d.decode([2, 5, 3]) == "alkylgroup"             # loc tokens vanish; look-ahead sees the loc token, not the next piece
    assert d.decode([3, 6]) == "group <cxsmi> "
    assert d.decode([11]) == "<other_99>"                  # unknown other token printed raw
    assert d.decode([4, 3]) == "group"                     # lone marker -> empty
    di = A.IdDecoder(toks, vocabulary, inverse, encode_index=True)
    assert di.decode([8, 9, 12, 12, 10, 8]) == "C C "      # <i> 7 7 </i> dropped
    assert d.batch_decode([[0, 6, 8, 7, 1, 0, 0], [0, 2, 3, 2, 3, 2, 1]]) == ["<cxsmi> C </cxsmi> ", "alkylgroup alkylgroup alkyl"]
    assert A.text_to_cxsmiles_opt("<markush><cxsmi> C C </cxsmi> <stable>x</stable>") == "CC"
    assert A.text_to_cxsmiles_opt("no tags") is None and A.text_to_cxsmiles_opt("<smi>C O</smi></s>", "ocsr") == "CO"


def test_word_boxes_and_cell_text_match_the_reference_functions():
    """f-3, pinned: split_bounding_box_for_words / prepare_cells_to_text (+ check_max_values, normalize_bbox_format,
    estimate_word_width, normalText) against outputs of the REFERENCE's own functions
    (ref: core/common/data_preprocessing.py:16-104, core/common/utils.py:204-222), executed unmodified by
    tools/make_golden_wordboxes.py with the same stand-in tokenizer.  Box coordinates: bit-exact floats."""
    import json
    from markushgrapher_amd import assembly as A
    from tools.make_golden_wordboxes import PieceTokenizer
    with open(os.path.join(GOLDEN, "host_wordboxes.json")) as f:
        g = json.load(f)
    tok = PieceTokenizer()
    for c in g["split_bounding_box_for_words"]:
        words, boxes = A.split_bounding_box_for_words(c["sentence"], c["bbox"], tok)
        assert words == c["words"]
        assert [list(b) for b in boxes] == c["boxes"]
    n_dropped = n_budget = 0
    for c in g["prepare_cells_to_text"]:
        words, boxes, tidx = A.prepare_cells_to_text(c["cells"], tok, **c["kwargs"])
        assert words == c["words"] and tidx == c["token_idx"]
        assert [list(b) for b in boxes] == c["boxes"]
        n_budget += tidx >= c["kwargs"]["max_sequence_length"] - 15
        total = sum(len([p for p in tok.tokenize(cell["text"]) if not p.isspace()]) for cell in c["cells"] if not cell["text"].isspace())
        n_dropped += len(words) < total
    assert n_dropped > 0 and n_budget > 0          # the >500 px drop and the token-budget stop are both exercised
    for b, r in g["misc"]["check_max_values"]:
        assert A.check_max_values(b) == r
    for b, r in g["misc"]["normalize_bbox_format"]:
        assert list(A.normalize_bbox_format(b, 512, 512)) == r
    for w_, r in g["misc"]["estimate_word_width"]:
        assert A.estimate_word_width(w_) == r
    for t, r in g["misc"]["normalText"]:
        assert A.normal_text(t) == r


def test_ocr_text_parser_matches_reference_outputs():
    """ocr_text.parse_ocr_string / clean_ocr_text against outputs of the reference's own functions (tools/make_golden_ocrtext.py)."""
    import json
    import os
    from markushgrapher_amd import ocr_text
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "host_ocrtext.json")))
    for c in g["parse"]:
        w, b = ocr_text.parse_ocr_string(c["in"])
        assert w == c["words"] and b == c["boxes"], (c["in"], w, b, c["words"], c["boxes"])
    for c in g["clean"]:
        assert ocr_text.clean_ocr_text(c["in"]) == c["out"], c["in"]
        assert ocr_text.clean_ocr_text(c["in"], end_tag=None) == c["out_no_end"], c["in"]


def test_id_decoder_matches_the_reference_method():
    """f-4, pinned: IdDecoder.decode against outputs of the REFERENCE's own MarkushTokenizer.decode_plus_decode_other_tokens
    (ref: core/common/markush_tokenizer.py:615-670), executed unmodified by tools/make_golden_idtext.py (rdkit / SmilesPE stubbed as
    empty modules - the method touches neither; stand-in id -> token table with every token class the method distinguishes).
    Both encode_index settings (ref: config/datasets/datasets_predict.yaml:8 uses True)."""
    import json
    from markushgrapher_amd import assembly as A
    with open(os.path.join(GOLDEN, "host_idtext.json")) as f:
        g = json.load(f)
    first, voc = g["first_other"], g["markush_vocabulary"]
    vocabulary = {v: f"<other_{first + i}>" for i, v in enumerate(voc)}
    inverse = {v: k for k, v in vocabulary.items()}
    dec = {ei: A.IdDecoder(g["tokens"], vocabulary, inverse, encode_index=ei) for ei in (False, True)}
    assert len(g["cases"]) >= 20
    for c in g["cases"]:
        assert dec[c["encode_index"]].decode(c["ids"]) == c["text"], (c["encode_index"], [g["tokens"][i] for i in c["ids"]], c["text"])


def test_text_to_cxsmiles_opt_pinned_on_the_reference_lines():
    """text_to_cxsmiles_opt against the outputs of the reference's OWN inline lines (utils/ocsr/utils_evaluation.py:306-352, exec'd
    unmodified by tools/make_golden_cxsmiles_opt.py with the names they use supplied by a harness): 15 texts x the three task names."""
    import json
    import os
    from markushgrapher_amd import assembly as A
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "host_cxsmiles_opt.json")))
    assert len(d["cases"]) == 45 and {c["task"] for c in d["cases"]} == {"ocsr", "ocxsr", "mdu"}
    for c in d["cases"]:
        assert A.text_to_cxsmiles_opt(c["text"], c["task"]) == c["opt"], c
    assert sum(c["opt"] is None for c in d["cases"]) >= 4          # mdu texts without a <cxsmi> span: the reference's except branch
