"""Mints tests/golden/pipeline_host.json: what the REFERENCE's own host code makes of the OCR strings of the configs[4] pipeline test,
executed unmodified in the build container.  Chain (ref: scripts/inference/inference.sh:165-184):
    OCR text --clean_ocr_text, parse_ocr_string (ocr/chemical_ocr.py:165-222, :438-446)--> cells
    --order_cells (core/datasets/mdu_dataset.py:78-80: sorted by (y0, x0))--> item {image, entities, cells, config}
    --encode_item (utils/common.py:14-97) = TaskCollator.collate (core/datasets/task_collator.py:28-107: prepare_cells_to_text,
      boxes / (w, h)) + processor(images, text=[instruction], text_pair=[words], boxes=[boxes])--> input_ids, bbox, attention_mask,
      pixel_values.
Loaded from /root/reference: chemical_ocr and utils.common by import, task_collator / data_preprocessing / utils by file path with
the stub modules of tools/make_golden_wordboxes.py.  The processor is STOCK UdopProcessor(LayoutLMv3ImageProcessor(apply_ocr=False,
size), UdopTokenizer) as begin.py:105-121 builds it (the fork's classes derive from these), the tokenizer vocabulary the stand-in of
tests/pipeline_fixture.py (the sentencepiece model is not available offline).  `order_cells` is a method of a dataset class that
imports cv2 / albumentations: its one line is restated here.  Only data is written.
    python tools/make_golden_pipeline.py
"""
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.make_golden_wordboxes import load_ref  # noqa: E402
from tests import pipeline_fixture as F  # noqa: E402

IMAGE_SIZE = 64           # the tiny main model's input size (synth.SHAPES["tiny"])


def main():
    from PIL import Image
    # (chemical_ocr, utils.common and the transformers classes first: the EMPTY torchvision stub that load_ref() registers afterwards
    #  would confuse `datasets` / transformers' availability probes)
    sys.path.insert(0, "/root/reference")
    spec = importlib.util.spec_from_file_location("ref_chemical_ocr", "/root/reference/markushgrapher/ocr/chemical_ocr.py")
    co = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(co)
    spec = importlib.util.spec_from_file_location("ref_utils_common", "/root/reference/markushgrapher/utils/common.py")
    uc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(uc)
    from transformers import LayoutLMv3ImageProcessor, UdopProcessor
    tok = F.make_udop_tokenizer()
    ip = LayoutLMv3ImageProcessor(apply_ocr=False, size={"height": IMAGE_SIZE, "width": IMAGE_SIZE}, image_mean=[0.5, 0.5, 0.5],
                                  image_std=[0.5, 0.5, 0.5])                                  # begin.py:105-109
    processor = UdopProcessor(image_processor=ip, tokenizer=tok)
    load_ref()                                        # registers markushgrapher.core.common.{utils, data_preprocessing} with the stubs
    spec = importlib.util.spec_from_file_location("markushgrapher.core.datasets.task_collator",
                                                  "/root/reference/markushgrapher/core/datasets/task_collator.py")
    tc = importlib.util.module_from_spec(spec)
    sys.modules["markushgrapher.core.datasets.task_collator"] = tc
    spec.loader.exec_module(tc)
    collator = tc.TaskCollator(tok)
    pages = F.pages_u8(len(F.OCR_TEXTS))
    cfg = {"normalize_bbox": True, "udop_tokenizer_only": True}          # config/datasets/datasets_predict.yaml:15 (labels are not used here)
    out = {"image_size": IMAGE_SIZE, "pages": []}
    for b, text in enumerate(F.OCR_TEXTS):
        words, boxes = co.parse_ocr_string(co.clean_ocr_text(text))                            # chemical_ocr.py:438-440
        cells = [{"bbox": bx, "text": w} for w, bx in zip(words, boxes)]                       # chemical_ocr.py:442-445
        cells = sorted(cells, key=lambda d: (d["bbox"][1], d["bbox"][0]))                      # mdu_dataset.py:78-80
        page = Image.fromarray(pages[b]).resize((IMAGE_SIZE, IMAGE_SIZE), resample=Image.LANCZOS)   # mdu_dataset.py:118 (512 there)
        item = {"image": page, "cells": cells, "config": cfg,
                "entities": {"question": F.QUESTION, "answer": "", "bbox": [0, 0, IMAGE_SIZE, IMAGE_SIZE]}}
        enc = uc.encode_item(item, processor, tok, None, collator, "test")
        pv = enc["pixel_values"].numpy()
        out["pages"].append({
            "ocr_text": text, "cells": cells,
            "input_ids": enc["input_ids"].tolist(), "bbox": [[float(x) for x in r] for r in enc["bbox"].tolist()],
            "attention_mask": enc["attention_mask"].tolist(),
            "pixel_sum": float(pv.astype(np.float64).sum()), "pixel_probe": [float(x) for x in pv[:, ::9, ::7].ravel()[:64]],
        })
        print(f"page {b}: {len(cells)} cells -> {len(out['pages'][-1]['input_ids'])} tokens; first boxes {out['pages'][-1]['bbox'][12:15]}")
    path = os.path.join(ROOT, "tests", "golden", "pipeline_host.json")
    with open(path, "w") as f:
        json.dump(out, f, ensure_ascii=False)
    print("wrote", path)


if __name__ == "__main__":
    main()
