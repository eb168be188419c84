"""Host-side input assembly and output post-processing around the hot path (SURVEY.md §8 "next" rows f-3 / f-4).

These are the reference's per-sample Python helpers restated so that a bs=32 batch can be built and decoded without
its dataset classes.  They are host code (string / list work, microseconds per image) - nothing here is on the GPU
path and nothing here imports the oracle.

Pinning status:
  * `DataCollator`            - pinned: fixtures minted by running the reference's own class, loaded by file path
                                (`tools/make_golden_host.py`), `tests/golden/host_collator.json`.
  * word boxes / cell text (f-3)  - pinned: `tests/golden/host_wordboxes.json` holds inputs and outputs of the reference's
                                own `split_bounding_box_for_words` / `prepare_cells_to_text`, executed from
                                /root/reference by `tools/make_golden_wordboxes.py` (a deterministic piece tokenizer stands
                                in for the sentencepiece model, which is not available offline).
  * id->text decoding (f-4)       - `IdDecoder` pinned: `tests/golden/host_idtext.json` holds inputs and outputs of the
                                reference's own `MarkushTokenizer.decode_plus_decode_other_tokens`, executed from
                                /root/reference by `tools/make_golden_idtext.py` (rdkit / SmilesPE stubbed as empty modules:
                                the method touches neither; stand-in id -> token table).  `text_to_cxsmiles_opt` restates a
                                few inline `str.replace` / `re.search` lines of `get_smiles_metrics`
                                (utils_evaluation.py:303-345) that are not callable on their own: known-answer tests.

Reference: markushgrapher/core/trainers/data_collator.py:11-108 (DataCollator, pad_sequence_native),
markushgrapher/core/common/data_preprocessing.py:11-104 (word boxes, prepare_cells_to_text),
markushgrapher/core/common/utils.py:212-222 (check_max_values, normalize_bbox_format),
markushgrapher/core/datasets/task_collator.py:26-107 (TaskCollator.collate),
markushgrapher/core/common/markush_tokenizer.py:607-670 (decode_plus_decode_other_tokens),
markushgrapher/utils/ocsr/utils_evaluation.py:286-345 (text -> CXSMILES string).
"""
from dataclasses import dataclass
import re
from typing import Optional

import torch

SP = "▁"      # sentencepiece word-start marker


# ---------------------------------------------------------------------------------------------------------------
# f-3: batching (data_collator.py:11-108)
# ---------------------------------------------------------------------------------------------------------------
def pad_sequence_native(seq, target_len, pad_value=0, dtype=torch.int):
    """Right-pad (or cut) a 1-D / [n,4] sequence to `target_len` rows (data_collator.py:11-20).  Lists become tensors
    of `dtype`; a tensor keeps its dtype and the padding takes it."""
    if not isinstance(seq, torch.Tensor):
        seq = torch.tensor(seq, dtype=dtype)
    n = seq.shape[0]
    if n >= target_len:
        return seq[:target_len]
    fill = torch.tensor([pad_value] * (target_len - n), dtype=seq.dtype)
    return torch.cat([seq, fill], dim=0)


_DECODER_KEYS = ("decoder_input_ids", "labels", "decoder_attention_mask", "decoder_seg_data")
_CHAR_KEYS = ("char_ids", "char_seg_data")
_STACK_KEYS = ("visual_seg_data", "definition_groups")


@dataclass
class DataCollator:
    """Batches per-sample feature dicts exactly like the reference collator: every sequence key is cut and padded to
    the FIXED `max_length` (decoder keys to `max_length_decoder`, char keys to min(longest, `max_length_char`)),
    pad value 0, [0,0,0,0] for `bbox`, -100 for `labels` / `image_mask_labels`; `pixel_values` are stacked; `image`
    is passed through from the last processed key (reference quirk: it re-emits the previous key's tensor)."""
    tokenizer: object = None
    padding: object = True
    max_length: Optional[int] = 1024
    max_length_decoder: Optional[int] = 512
    max_length_char: Optional[int] = 1024 + 512
    pad_to_multiple_of: Optional[int] = None

    def __call__(self, features):
        if features[0] is None:
            return {"placeholder": torch.zeros(size=(2, 2), dtype=torch.long)}
        first = features[0]
        char_len = None
        if "char_ids" in first:
            char_len = min(max(int(f["char_ids"].shape[0]) for f in features), self.max_length_char)
        batch = {}
        prev = None
        for key in first.keys():
            if key == "pixel_values":
                continue
            pad = [0] * 4 if key == "bbox" else (-100 if key in ("labels", "image_mask_labels") else 0)
            if key in _DECODER_KEYS:
                tgt = self.max_length_decoder
            elif key in _CHAR_KEYS:
                tgt = char_len
            elif key in _STACK_KEYS or key == "image":
                tgt = None
            else:
                tgt = self.max_length
            if key == "image":
                out = prev          # the reference leaves `batched_feature` untouched for this key
            elif tgt is None:
                out = torch.stack([f[key] for f in features], dim=0)
            else:
                if key not in _CHAR_KEYS:
                    for f in features:           # the reference truncates the caller's features in place
                        f[key] = f[key][:tgt]
                out = torch.stack([pad_sequence_native(f[key], tgt, pad) for f in features], dim=0)
            batch[key] = out
            prev = out
        if "pixel_values" in first:
            batch["pixel_values"] = torch.stack([f["pixel_values"] for f in features])
        return batch


def collate_for_generate(features, pad_to=None):
    """Inference-side batching for `MarkushgrapherForConditionalGeneration.generate`: pads `input_ids` (0), `bbox`
    (0-box) and `attention_mask` (0) to the longest sample (or `pad_to`) instead of the trainer's fixed 1024 - the
    encoder cost is linear in the padded length - and stacks `pixel_values`.  Same pad values as `DataCollator`."""
    L = max(int(f["input_ids"].shape[0]) for f in features) if pad_to is None else int(pad_to)
    out = {}
    for key, pad in (("input_ids", 0), ("bbox", [0] * 4), ("attention_mask", 0)):
        if key in features[0]:
            out[key] = torch.stack([pad_sequence_native(f[key], L, pad) for f in features], dim=0)
    if "attention_mask" not in out:
        lens = torch.tensor([min(int(f["input_ids"].shape[0]), L) for f in features])
        out["attention_mask"] = (torch.arange(L)[None, :] < lens[:, None]).to(torch.long)
    if "pixel_values" in features[0]:
        out["pixel_values"] = torch.stack([f["pixel_values"] for f in features])
    return out


# ---------------------------------------------------------------------------------------------------------------
# f-3: OCR cells -> words + per-word boxes (data_preprocessing.py:11-104, task_collator.py:26-107)  [pinned: host_wordboxes.json]
# ---------------------------------------------------------------------------------------------------------------
def estimate_word_width(piece):
    """12 px per character, the word-start marker not counted; a lone marker counts as one character."""
    n = 1 if piece == SP else sum(1 for c in piece if c != SP)
    return 12 * n


def split_bounding_box_for_words(sentence, bounding_box, tokenizer):
    """Cuts a cell box into one box per sentencepiece piece, widths proportional to the estimated piece widths, laid
    out left to right with a running left edge (so rounding accumulates exactly like the reference)."""
    pieces = tokenizer.tokenize(sentence)
    widths = [estimate_word_width(p) for p in pieces]
    total = sum(widths)
    x0, y0, x1, y1 = bounding_box
    boxes, left = [], x0
    for wd in widths:
        step = (x1 - x0) * (wd / total)
        boxes.append((left, y0, left + step, y1))
        left += step
    return pieces, boxes


def normal_text(t):
    if type(t) is float and t == int(t):
        t = int(t)
    return str(t).strip()


def check_max_values(box, max_value=500):
    return any(c > max_value for c in box)


def normalize_bbox_format(box, w, h):
    return tuple(int((v / s) * 500) for v, s in zip(box, (w, h, w, h)))


def prepare_cells_to_text(cells, tokenizer, w, h, normalize_bbox, max_sequence_length=512):
    """Cells (text + box in [0,1]) -> pieces and pixel boxes; whitespace cells and pieces are skipped, boxes with a
    coordinate > 500 are dropped, the scan stops 15 tokens short of `max_sequence_length` (inner loop) or at it."""
    words, boxes, n_tok = [], [], 0
    for cell in cells:
        text = cell["text"]
        if text.isspace():
            continue
        bx = cell["bbox"]
        pieces, pboxes = split_bounding_box_for_words(text, [bx[0] * w, bx[1] * h, bx[2] * w, bx[3] * h], tokenizer)
        for piece, pb in zip(pieces, pboxes):
            if piece.isspace():
                continue
            if not normalize_bbox:
                pb = normalize_bbox_format(pb, w, h)
            if check_max_values(pb):
                continue
            words.append(normal_text(piece))
            boxes.append(pb)
            n_tok += len(tokenizer.tokenize(normal_text(piece)))
            if n_tok >= max_sequence_length - 15:
                break
        if n_tok >= max_sequence_length:
            break
    return words, boxes, n_tok


def collate_item(item, tokenizer, normalize_bbox):
    """TaskCollator.collate: (image, "Question Answering. <question>", pieces, boxes in [0,1] if normalize_bbox,
    [answer, "</s>"])."""
    image = item["image"]
    w, h = image.size
    words, boxes, _ = prepare_cells_to_text(item["cells"], tokenizer, w, h, normalize_bbox)
    if normalize_bbox:
        boxes = [[b[0] / w, b[1] / h, b[2] / w, b[3] / h] for b in boxes]
    ent = item["entities"]
    return image, f"Question Answering. {ent['question']}", words, boxes, [normal_text(ent["answer"]), "</s>"]


# ---------------------------------------------------------------------------------------------------------------
# f-4: generated ids -> text -> CXSMILES string (markush_tokenizer.py:615-670 [pinned], utils_evaluation.py:286-345)
# ---------------------------------------------------------------------------------------------------------------
class IdDecoder:
    """Table-driven form of `decode_plus_decode_other_tokens`.  Every vocabulary id is classified once:
    location tokens are dropped, `<other_N>` tokens map to their Markush vocabulary string + " ", ordinary pieces lose
    the leading marker; whether a piece is followed by a space depends only on a per-id flag of the NEXT token
    (marker anywhere in it, or "other" in it).  Decoding a sequence is then two table lookups per token."""

    def __init__(self, id_to_token, vocabulary, vocabulary_inverse, encode_index=False):
        self.encode_index = bool(encode_index)
        i_open = vocabulary.get("<i>") if self.encode_index else None
        i_close = vocabulary.get("</i>") if self.encode_index else None
        n = len(id_to_token)
        self.text = [""] * n          # emitted text (without the look-ahead space)
        self.kind = [0] * n           # 0 ordinary piece, 1 emitted verbatim (other / unknown other), 2 dropped
        self.opens = [False] * n
        self.closes = [False] * n
        self.spaces_prev = [False] * n
        self.close_exact = [False] * n
        for i, tok in enumerate(id_to_token):
            self.close_exact[i] = i_close is not None and tok == i_close
            self.spaces_prev[i] = (SP in tok) or ("other" in tok)
            self.opens[i] = i_open is not None and i_open in tok
            self.closes[i] = i_close is not None and i_close in tok
            angled = "<" in tok and ">" in tok
            if "loc" in tok and angled:
                self.kind[i] = 2
            elif "other" in tok and angled:
                self.kind[i] = 1
                self.text[i] = vocabulary_inverse[tok] + " " if tok in vocabulary_inverse else tok
            else:
                self.text[i] = tok[1:] if tok[:1] == SP else tok

    def decode(self, ids):
        ids = [int(i) for i in ids]
        out, skipping = [], False
        for p, i in enumerate(ids):
            if skipping:
                # inside <i> ... </i>: everything up to the closing token is dropped; the closing token itself is
                # exactly `</i>`'s vocabulary string in the reference's test (`token != vocabulary["</i>"]`)
                if not self.close_exact[i]:
                    continue
            skipping = False
            if self.opens[i]:
                skipping = True
                continue
            if self.closes[i]:
                continue
            k = self.kind[i]
            if k == 2:
                continue
            if k == 1:
                out.append(self.text[i])
            else:
                nxt = ids[p + 1] if p + 1 < len(ids) else None
                out.append(self.text[i] + (" " if nxt is not None and self.spaces_prev[nxt] else ""))
        return "".join(out)

    def batch_decode(self, ids_2d, eos_id=1, pad_id=0):
        """Per row: the reference decodes `predicted_ids[0][1:-1]` of a bs-1 generate() (start token and final EOS
        removed).  In a batch, rows are padded after their EOS, so the cut is made at the first EOS instead."""
        texts = []
        for row in ids_2d:
            row = [int(v) for v in row][1:]
            if eos_id in row:
                row = row[:row.index(eos_id)]
            else:
                row = row[:-1]
            texts.append(self.decode(row))
        return texts


_CXSMI = re.compile(re.escape("<cxsmi>") + r"(.*?)" + re.escape("</cxsmi>"))


def text_to_cxsmiles_opt(text, task="mdu"):
    """utils_evaluation.py:306-352: strip the task tags, `</s>` and spaces; for "mdu" only the first
    <cxsmi>...</cxsmi> span is kept (None when absent).  Pinned on the outputs of those reference lines themselves
    (tools/make_golden_cxsmiles_opt.py -> tests/golden/host_cxsmiles_opt.json)."""
    if task == "ocsr":
        return text.replace("<smi>", "").replace("</smi>", "").replace("</s>", "").replace(" ", "")
    if task == "mdu":
        m = _CXSMI.search(text)
        if m is None:
            return None
        text = "<cxsmi>" + m.group(1) + "</cxsmi>"
    return text.replace("<cxsmi>", "").replace("</cxsmi>", "").replace("</s>", "").replace(" ", "")
