"""BASELINE.json configs[4] as one path (markushgrapher_amd/pipeline.py): pages -> device preprocessing -> ChemicalOCR generate -> text ->
cells -> VTL inputs -> VTL generate.  A tiny OCR model with SCRIPTED weights (tests/pipeline_fixture.py) deterministically 'reads' a fixed
cell string per page; everything downstream of that string is compared with tests/golden/pipeline_host.json, which holds what the
REFERENCE's own host code (parse_ocr_string, clean_ocr_text, TaskCollator.collate, encode_item with the stock UDOP processor) makes of
the same strings (tools/make_golden_pipeline.py).  `emu`: both engines on the CPU SIMT emulator; `hip`: MI355X."""
import json
import os

import numpy as np
import pytest

from markushgrapher_amd import synth
from markushgrapher_amd.pipeline import Configs4Pipeline, cells_from_ocr_text, encode_cells, order_cells
from tests import pipeline_fixture as F
from tests.backends import get_backend, make_engine, NumpyMem
from tests.conftest import GOLDEN, load_golden
from tests.test_oracle_golden import _weights

BACKENDS = [pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)]


def _golden():
    with open(os.path.join(GOLDEN, "pipeline_host.json")) as f:
        return json.load(f)


def test_host_chain_matches_the_reference_outputs():
    """OCR string -> cells -> input_ids / bbox: the build's restatements + the stock tokenizer against the reference's own chain."""
    g = _golden()
    tok = F.make_udop_tokenizer()
    assert [p["ocr_text"] for p in g["pages"]] == F.OCR_TEXTS
    for p in g["pages"]:
        cells = cells_from_ocr_text(p["ocr_text"])
        assert order_cells(cells) == p["cells"]
        ids, bb = encode_cells(cells, tok, g["image_size"])
        assert ids.tolist() == p["input_ids"]
        assert np.array_equal(bb, np.asarray(p["bbox"], np.float32))
    assert any(c["bbox"][1] > n["bbox"][1] for p in g["pages"] for c, n in zip(cells_from_ocr_text(p["ocr_text"]), cells_from_ocr_text(p["ocr_text"])[1:]))


def _engines(be_name):
    from markushgrapher_amd.ocr import OcrEngine
    g3 = load_golden("g3_trained_tiny.npz")
    shape, sd = _weights(g3)
    main = make_engine(be_name, shape, sd)
    s = F.scripted_ocr_shape()
    be = get_backend(be_name)
    ocr = OcrEngine(s, lib=be.lib, mem=NumpyMem()) if be_name == "emu" else OcrEngine(s)
    ocr.load_state_dict(F.scripted_ocr_state_dict())
    return main, ocr, shape, s


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("continuous", [False, True])
def test_pages_to_ids_in_one_path(be_name, continuous):
    # continuous: BOTH stages in their queue forms (OCR: 2 decode rows for the 4 pages; VTL: the continuous decoder)
    g = _golden()
    main, ocr, shape, s = _engines(be_name)
    id_to_piece, chains, starts = F.ocr_vocab_and_chains()
    pipe = Configs4Pipeline(ocr, main, F.make_udop_tokenizer(), lambda row: F.detokenize(id_to_piece, row, s.eos_token_id, s.pad_token_id),
                            F.ocr_prompts(), ocr_max_new_tokens=64, max_length=16, continuous=continuous, ocr_slots=2 if continuous else 0)
    pages = F.pages_u8(len(F.OCR_TEXTS))
    res = pipe(pages)
    # stage 1: the scripted OCR model walked its chains (ids), the stand-in detokeniser gives the designed strings
    for b, chain in enumerate(chains):
        assert res.ocr_new_ids[b, :len(chain)].tolist() == chain, b
    assert res.ocr_texts == F.OCR_TEXTS
    # in between: cells and VTL inputs equal the reference's
    L = res.input_ids.shape[1]
    assert L == max(len(p["input_ids"]) for p in g["pages"])
    for b, p in enumerate(g["pages"]):
        n = len(p["input_ids"])
        assert order_cells(res.cells[b]) == p["cells"]
        assert res.input_ids[b, :n].tolist() == p["input_ids"] and np.all(res.input_ids[b, n:] == 0)
        assert np.array_equal(res.bbox[b, :n], np.asarray(p["bbox"], np.float32)) and np.all(res.bbox[b, n:] == 0)
        assert res.attention_mask[b].tolist() == [1] * n + [0] * (L - n)
    # the device preprocessing is the reference's pixel path (PIL LANCZOS + 1/255 + 0.5/0.5): checksums of its pixel_values
    pix = main.mem.numpy(main.preprocess(pages))
    for b, p in enumerate(g["pages"]):
        assert abs(float(pix[b].astype(np.float64).sum()) - p["pixel_sum"]) < 1e-3
        assert np.array_equal(pix[b][:, ::9, ::7].ravel()[:64], np.asarray(p["pixel_probe"], np.float32))
    # stage 2: the ids are what the VTL engine returns for the reference-made inputs (same engine, same pixel values)
    want, _, _ = main.generate(res.input_ids, res.bbox, res.attention_mask, pix, max_length=16)
    want = main.mem.numpy(want)
    assert res.ids.shape == want.shape and np.array_equal(res.ids, want)
    assert np.all(res.ids[:, 0] == shape.decoder_start_token_id) and len({tuple(r) for r in res.ids.tolist()}) > 1


@pytest.mark.parametrize("be_name", BACKENDS)
def test_per_image_padding_semantics(be_name):
    """mg_set_padding_semantics(1): in a padded batch every image is computed as if it were alone and unpadded - what the reference's
    batch-size-1 loop computes (ref: utils/ocsr/utils_evaluation.py:140).  Stock HF batched semantics (the default, what the golden
    fixtures of padded batches pin) leaves the padded text slots between the text and the patches, where UDOP's 1-D position bias
    counts them: there an image's result depends on the padding of its batch.  Here: the 4 pages of the pipeline fixture (14 - 26
    tokens) as one padded batch against each page alone: greedy ids equal, attended encoder rows equal to accumulation-order noise."""
    g = _golden()
    main, _, shape, _ = _engines(be_name)
    pages = F.pages_u8(len(F.OCR_TEXTS))
    pix = main.mem.numpy(main.preprocess(pages)).copy()
    feats = [(np.asarray(p["input_ids"], np.int64), np.asarray(p["bbox"], np.float32)) for p in g["pages"]]
    L = max(len(i) for i, _ in feats)
    ids = np.zeros((4, L), np.int64); bb = np.zeros((4, L, 4), np.float32); am = np.zeros((4, L), np.int64)
    for b, (i, x) in enumerate(feats):
        ids[b, :len(i)] = i; bb[b, :len(i)] = x; am[b, :len(i)] = 1
    alone_ids, alone_enc = [], []
    for b, (i, x) in enumerate(feats):
        one = (i[None], x[None], np.ones((1, len(i)), np.int64), pix[b:b + 1])
        o, _, _ = main.generate(*one, max_length=16)
        alone_ids.append(main.mem.numpy(o)[0].copy())
        e, _ = main.encode(*one)
        alone_enc.append(main.mem.numpy(e)[0].copy())
    assert main.set_padding_semantics(True) is False
    try:
        out, _, _ = main.generate(ids, bb, am, pix, max_length=16)
        out = main.mem.numpy(out).copy()
        enc, msk = main.encode(ids, bb, am, pix)
        enc, msk = main.mem.numpy(enc).copy(), main.mem.numpy(msk).copy()
        outs, lens, _ = main.generate_stream(ids, bb, am, pix, max_length=16, chunk=2, slots=3, pool_chunks=2)
        outs, lens = main.mem.numpy(outs), main.mem.numpy(lens)
    finally:
        assert main.set_padding_semantics(False) is True
    P = shape.num_patches
    for b, (i, _) in enumerate(feats):
        n = len(alone_ids[b])
        assert np.array_equal(out[b, :n], alone_ids[b]) and np.all(out[b, n:] == shape.pad_token_id), b
        assert np.array_equal(outs[b, :lens[b]], alone_ids[b][:lens[b]])
        Lb = len(i)
        # output contract [text L | patches P]: text rows, then (at offset L) the patch rows; mask of the padded text slots 0
        assert msk[b, :Lb].all() and not msk[b, Lb:L].any()
        a = np.concatenate([enc[b, :Lb], enc[b, L:L + P]]); r = np.concatenate([alone_enc[b][:Lb], alone_enc[b][Lb:Lb + P]])
        keep = np.concatenate([msk[b, :Lb], msk[b, L:L + P]]).astype(bool)
        assert np.abs(a - r)[keep].max() < 2e-3, (b, float(np.abs(a - r)[keep].max()))
    # and the default (stock batched) semantics really is a different computation for the padded pages
    enc0, _ = main.encode(ids, bb, am, pix)
    enc0 = main.mem.numpy(enc0)
    short = int(np.argmin([len(i) for i, _ in feats]))
    Ls = len(feats[short][0])
    assert np.abs(enc0[short, :Ls] - alone_enc[short][:Ls]).max() > 10 * 2e-3


@pytest.mark.parametrize("be_name,continuous,reps", [
    ("emu", False, 2), pytest.param("hip", False, 3, marks=pytest.mark.gpu), pytest.param("hip", True, 3, marks=pytest.mark.gpu)])
def test_contexts_in_flight_inside_the_stages_equal_one_context(be_name, continuous, reps):
    """main_inflight / ocr_inflight: the VTL stage's batches over execution contexts (host stage pipelined with them), the OCR stage's
    pages over contexts of the OCR model - page for page the strings, VTL inputs and ids of the one-context call (a VTL batch size that
    does not divide the page count, an uneven OCR split; `emu` runs the contexts' device work one at a time)."""
    main, ocr, shape, s = _engines(be_name)
    id_to_piece, chains, starts = F.ocr_vocab_and_chains()
    n = len(F.OCR_TEXTS)
    prompts = np.concatenate([F.ocr_prompts()] * reps, axis=0)
    pipe = Configs4Pipeline(ocr, main, F.make_udop_tokenizer(), lambda row: F.detokenize(id_to_piece, row, s.eos_token_id, s.pad_token_id),
                            prompts, ocr_max_new_tokens=64, max_length=16, continuous=continuous, ocr_slots=2 if continuous else 0, main_batch=3)
    pages = np.concatenate([F.pages_u8(n)] * reps, axis=0)
    if be_name == "hip":
        import torch
        pages = torch.from_numpy(pages).cuda()
    try:
        want = pipe(pages)
        pipe.main_inflight, pipe.ocr_inflight = 2, 3          # (the emulator backend keeps the OCR stage on one context)
        for _ in range(1 if be_name == "emu" else 2):          # second call: warm contexts (replayed graphs)
            got = pipe(pages)
            assert got.ocr_texts == want.ocr_texts == F.OCR_TEXTS * reps
            assert got.cells == want.cells
            L = min(got.input_ids.shape[1], want.input_ids.shape[1])
            assert np.array_equal(got.input_ids[:, :L], want.input_ids[:, :L]) and np.array_equal(got.attention_mask[:, :L], want.attention_mask[:, :L])
            assert np.array_equal(got.bbox[:, :L], want.bbox[:, :L])
            W = min(got.ids.shape[1], want.ids.shape[1])
            assert np.array_equal(got.ids[:, :W], want.ids[:, :W])
            assert np.all(got.ids[:, W:] == shape.pad_token_id) and np.all(want.ids[:, W:] == shape.pad_token_id)
            n_new = min(got.ocr_new_ids.shape[1], want.ocr_new_ids.shape[1])
            assert np.array_equal(got.ocr_new_ids[:, :n_new], want.ocr_new_ids[:, :n_new])
    finally:
        pipe.close()
        main.set_padding_semantics(False)
