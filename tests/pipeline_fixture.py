"""Fixture of the configs[4] pipeline test (tests/test_pipeline.py, tools/make_golden_pipeline.py): everything that is NOT the code
under test and has no offline original.

* `UDOP_STANDIN_VOCAB`: a Unigram vocabulary (piece, score) for stock `UdopTokenizer(vocab=...)` - the reference's sentencepiece model
  (t5 / UDOP, 33k pieces) is not available offline; the stand-in has ids < 500 so that the tiny main model (vocab 500) can embed them.
* `OCR_PIECES` + `scripted_ocr_state_dict`: a tiny Idefics3-shaped OCR model whose weights make greedy decoding walk a scripted chain
  of tokens (lm_head row of a token's successor = that token's embedding direction, embeddings dominate the residual stream), so that
  page b deterministically 'reads' the cell string OCR_TEXTS[b] whatever the pixels are: the pipeline test needs a known OCR output,
  not OCR quality.  The page's chain is chosen by the last prompt token.  The id -> piece table is the stand-in OCR tokenizer.
"""
import dataclasses

import numpy as np

from markushgrapher_amd import synth
from markushgrapher_amd.ocr_shapes import PRESETS, recipe_state_dict

# ---------------------------------------------------------------------------------------------------------------------------
# main model input tokenizer (stand-in for the UDOP sentencepiece model)
# ---------------------------------------------------------------------------------------------------------------------------
from markushgrapher_amd.standin import make_udop_tokenizer, udop_standin_vocab  # noqa: E402,F401  (the stand-in lives with the synthetic inputs)


QUESTION = "What markush structure is in the image?"           # ref: mdu_dataset.py:120-124

# ---------------------------------------------------------------------------------------------------------------------------
# scripted OCR stage
# ---------------------------------------------------------------------------------------------------------------------------
# what each page 'reads' (current grammar `[page box>]x1>y1>x2>y2>text`, chemical_ocr.py:165-199); cells deliberately out of reading
# order, one cell whose box leaves the 500-px window (dropped by the reference), one whitespace-only text
OCR_TEXTS = [
    "<ocr>0>0>500>500>260>300>420>330>R1 = alkyl group\n40>60>200>90>wherein R2 is OH\n40>20>120>50>Cl</ocr>",
    "<ocr>10>400>90>430>Ar = phenyl or Het\n300>40>360>70>NH\n120>40>200>70>Me</ocr>",
    "<ocr>0>0>500>500>30>30>130>60>R3 and R4 may be the same\n30>100>230>130>each represents a hydrogen atom\n400>480>505>499>Br</ocr>",
    "<ocr>200>200>300>230>Et</ocr>",
]
OCR_PROMPT_TEXT = 10          # text tokens of the prompt (around the <image> block)


def ocr_vocab_and_chains():
    """-> (id_to_piece, chains, starts) of the scripted tiny OCR model (markushgrapher_amd.ocr_shapes.script_texts)."""
    from markushgrapher_amd.ocr_shapes import script_texts
    return script_texts(scripted_ocr_shape(), OCR_TEXTS, 3)


def scripted_ocr_shape():
    return dataclasses.replace(PRESETS["tiny"])


def scripted_ocr_state_dict():
    from markushgrapher_amd.ocr_shapes import scripted_state_dict
    _, chains, starts = ocr_vocab_and_chains()
    return scripted_state_dict(scripted_ocr_shape(), chains, starts, gain=0.5, embed_scale=48.0)


def ocr_prompts():
    from markushgrapher_amd.ocr_shapes import scripted_prompts
    _, chains, starts = ocr_vocab_and_chains()
    return scripted_prompts(scripted_ocr_shape(), chains, starts, OCR_PROMPT_TEXT)


def detokenize(id_to_piece, row, eos_id, pad_id):
    from markushgrapher_amd.ocr_shapes import detokenize as d
    return d(id_to_piece, row, eos_id, pad_id)


def pages_u8(n, size=128, seed=5):
    return synth.synth_pages_u8(n, size, seed)
