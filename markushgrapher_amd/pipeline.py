"""BASELINE.json configs[4] as ONE path: page crops -> ChemicalOCR (prefill + greedy decode) -> OCR cells -> VTL inputs -> VTL encoder +
CXSMILES decoder -> token ids (-> text).  What the reference runs as two processes with a dataset on disk in between
(ref: scripts/inference/inference.sh:165-184: image_dir_to_hf_dataset.py --apply_ocr, then eval.py) happens here in one loop on one GPU,
the intermediate staying in memory:

    pages u8 [B,H,W,3] (device)
      -> mg_preprocess_pages                    Pillow-exact LANCZOS to 512 px, x/255, mean = std = 0.5            (both models read it:
                                                Idefics3ImageProcessor resamples with LANCZOS and normalises with 0.5 / 0.5 too, and a
                                                512-px page is one frame at max_image_size 512)
      -> OcrEngine.generate                     ref: ocr/chemical_ocr.py:366-392
      -> detokenise, clean_ocr_text, parse_ocr_string -> cells            ref: chemical_ocr.py:386-390, 438-446     (ocr_text.py)
      -> order_cells, TaskCollator.collate, tokenizer(text, text_pair, boxes)   ref: mdu_dataset.py:78-80, 210; task_collator.py:28-107;
                                                                                utils/common.py:34-42               (assembly.py)
      -> collate_for_generate (pad to the longest of the batch)
      -> Engine.generate / generate_stream      ref: utils/ocsr/utils_evaluation.py:269-285
      -> ids (-> IdDecoder text, optional)

Host work (string parsing, tokenising ~20-400 pieces per page) is the stock tokenizer's and a few list operations per page; it is not
accelerated.  There is no CPU fallback for the two model stages: both engines raise without the HIP library.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

from . import assembly, ocr_text

QUESTION = "What markush structure is in the image?"            # ref: core/datasets/mdu_dataset.py:120-124


def order_cells(cells):
    """ref: mdu_dataset.py:78-80 - reading order by (y0, x0)."""
    return sorted(cells, key=lambda d: (d["bbox"][1], d["bbox"][0]))


def cells_from_ocr_text(text: str):
    """ref: chemical_ocr.py:438-446."""
    words, boxes = ocr_text.parse_ocr_string(ocr_text.clean_ocr_text(text))
    return [{"bbox": b, "text": w} for w, b in zip(words, boxes)]


class _Size:
    def __init__(self, w, h):
        self.size = (w, h)


class _MemoTokenizer:
    """The word-box splitter asks the tokenizer for the pieces of every cell text and again for every piece (token budget,
    ref: data_preprocessing.py:59-104); OCR pages repeat the same few hundred strings.  `tokenize` results are memoised per string -
    same pieces, a dictionary lookup instead of a tokenizer call; every other attribute / call goes to the wrapped tokenizer."""

    def __init__(self, tok):
        self._tok, self._memo = tok, {}

    def tokenize(self, text):
        r = self._memo.get(text)
        if r is None:
            r = self._memo[text] = self._tok.tokenize(text)
            if len(self._memo) > 200000:
                self._memo.clear()
        return list(r)

    def __call__(self, *a, **k):
        return self._tok(*a, **k)

    def __getattr__(self, name):
        return getattr(self._tok, name)


def encode_cells(cells, tokenizer, image_size: int, question: str = QUESTION, normalize_bbox: bool = True):
    """One page's OCR cells -> (input_ids [L] int64, bbox [L,4] float32): `encode_item` without the pixel part (ref: utils/common.py:
    14-42): TaskCollator.collate's words / boxes, then the tokenizer call the processor makes."""
    item = {"image": _Size(image_size, image_size), "cells": order_cells(cells), "entities": {"question": question, "answer": ""}}
    _, instruction, words, boxes, _ = assembly.collate_item(item, tokenizer, normalize_bbox)
    enc = tokenizer(text=[instruction], text_pair=[words], boxes=[[list(map(float, b)) for b in boxes]], return_tensors="np", padding=False,
                    truncation=False)
    return np.asarray(enc["input_ids"][0], np.int64), np.asarray(enc["bbox"][0], np.float32)


@dataclass
class PipelineResult:
    ids: np.ndarray                   # [B, T] decoder ids (start token first, pad after EOS)
    ocr_new_ids: np.ndarray           # [B, n] what the OCR stage generated
    ocr_texts: List[str]
    cells: List[list]
    input_ids: np.ndarray             # [B, L] what the VTL model read
    bbox: np.ndarray
    attention_mask: np.ndarray
    timings: dict = field(default_factory=dict)


class Configs4Pipeline:
    """ocr: OcrEngine; main: Engine; tokenizer: the main model's tokenizer (UdopTokenizer-compatible call); ocr_detok: new-token ids of
    one page -> text (the OCR processor's batch_decode); ocr_prompt_ids [L] or [B, L]: the tokenised chat prompt with the <image>
    block (equal length for every page: one prompt, one 512-px frame)."""

    def __init__(self, ocr, main, tokenizer, ocr_detok: Callable, ocr_prompt_ids, ocr_max_new_tokens: int = 4096, question: str = QUESTION,
                 max_length: int = 512, min_length: int = 0, num_beams: int = 1, continuous: bool = False, main_batch: int = 32,
                 ocr_slots: int = 0, per_image_padding: bool = True, main_inflight: int = 1, ocr_inflight: int = 1, overlap_slab: int = 0):
        self.ocr, self.main, self.tokenizer, self.ocr_detok = ocr, main, _MemoTokenizer(tokenizer), ocr_detok
        self.ocr_prompt_ids = np.asarray(ocr_prompt_ids, np.int64)
        self.ocr_max_new_tokens, self.question = int(ocr_max_new_tokens), question
        self.max_length, self.min_length, self.num_beams, self.continuous = int(max_length), int(min_length), int(num_beams), bool(continuous)
        self.main_batch = int(main_batch)      # pages per VTL call (the OCR stage may take more pages per call: its model is 6x smaller)
        self.ocr_slots = int(ocr_slots)        # > 0: the OCR stage's queue form (mg_ocr_generate_stream) with that many decode rows
        # > 1: the VTL stage keeps that many batches of `main_batch` pages in flight (execution contexts, markushgrapher_amd/inflight.py)
        self.main_inflight = int(main_inflight)
        self._fl = None
        # > 1: the OCR stage splits its pages over that many execution contexts of the OCR model (mg_ocr_clone), a thread + stream each
        self.ocr_inflight = int(ocr_inflight)
        self._ocr_ctx = []
        # > 0 (with main_inflight > 1): the two GPU stages OVERLAP - the pages go through the OCR stage in slabs of that many pages on the OCR
        # contexts' OWN streams while the VTL contexts decode the previous slab.  The OCR decode step is a chain of ~120 latency-sized
        # launches (0.06 of the HBM peak), the VTL stage is bandwidth- / matrix-bound: each fills what the other leaves idle.
        self.overlap_slab = int(overlap_slab)
        self._ocr_streams = None
        # pages of one batch have different token counts; with per-image padding semantics every page is computed as the reference
        # computes it (alone, unpadded: its batch size is 1), whatever the batch was padded to (mg_set_padding_semantics)
        # (the engine's previous setting is put back by close(): other users of `main` keep stock batched semantics)
        self._prev_padding = self.main.set_padding_semantics(True) if per_image_padding else None
        if ocr.shape.image_size != main.shape.image_size:
            raise ValueError("the two stages share the preprocessed page: equal input sizes expected (512 px in the reference)")

    # ---- the three stages --------------------------------------------------------------------------------------------------
    def _ocr_part(self, ocr, pix, prompt):
        B = int(pix.shape[0])
        if self.ocr_slots > 0:
            new, _, steps = ocr.generate_stream(prompt, pix[:, None], self.ocr_max_new_tokens, slots=min(self.ocr_slots, B), chunk=min(B, 128))
        else:
            (new, _), steps = ocr.generate(prompt, pix[:, None], self.ocr_max_new_tokens), None
        return (new.cpu().numpy() if hasattr(new, "cpu") else np.asarray(new)), steps

    def stage_ocr(self, pages_u8, page0: int = 0):
        """-> (pix [B,3,I,I] device, new ids [B,n] numpy, OCR decode steps or None).  page0: index of the first page in the caller's
        list (per-page prompts are taken from ocr_prompt_ids[page0 : page0 + B])."""
        pix = self.main.preprocess(pages_u8)                                  # [B, 3, I, I] f32 on the device, read by both stages
        B = int(pix.shape[0])
        prompt = self.ocr_prompt_ids[page0:page0 + B] if self.ocr_prompt_ids.ndim == 2 else np.repeat(self.ocr_prompt_ids[None], B, axis=0)
        n = min(self.ocr_inflight, B)
        torch = getattr(self.main.mem, "torch", None)
        if n <= 1 or torch is None or not torch.cuda.is_available():          # (the CPU emulator of the tests runs one context)
            new, steps = self._ocr_part(self.ocr, pix, prompt)
            return pix, new, steps
        import threading
        from .inflight import shared_streams
        if self.overlap_slab > 0:                                             # stages overlap: the OCR contexts keep streams of their own
            if self._ocr_streams is None or len(self._ocr_streams) < n:
                self._ocr_streams = [torch.cuda.Stream(self.main.mem.device) for _ in range(n)]
            streams = self._ocr_streams
        else:
            streams = shared_streams(torch, self.main.mem.device, n)         # the VTL contexts' streams (the stages alternate)
        while len(self._ocr_ctx) < n:
            self._ocr_ctx.append((self.ocr if not self._ocr_ctx else self.ocr.clone(), streams[len(self._ocr_ctx)]))
        torch.cuda.current_stream().synchronize()                             # pix is complete before the other streams read it
        per = -(-B // n)
        parts, errs = [None] * n, []

        def work(i):
            try:
                ctx, st = self._ocr_ctx[i]
                with torch.cuda.device(st.device), torch.cuda.stream(st):
                    parts[i] = self._ocr_part(ctx, pix[i * per:(i + 1) * per], prompt[i * per:(i + 1) * per])
                    st.synchronize()
            except BaseException as e:
                errs.append(e)
        th = [threading.Thread(target=work, args=(i,)) for i in range(n) if i * per < B]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        parts = [p for p in parts if p is not None]
        width = max(p[0].shape[1] for p in parts)
        new = np.concatenate([np.pad(p[0], ((0, 0), (0, width - p[0].shape[1])), constant_values=self.ocr.shape.pad_token_id) for p in parts], axis=0)
        return pix, new, max(p[1] or 0 for p in parts) or None

    def stage_host(self, new):
        """OCR ids -> (texts, cells, input_ids, bbox, attention_mask): string and tokenizer work, no GPU."""
        import torch
        texts = [self.ocr_detok(row) for row in new]
        cells = [cells_from_ocr_text(x) for x in texts]
        I = self.main.shape.image_size
        items = [assembly.collate_item({"image": _Size(I, I), "cells": order_cells(c), "entities": {"question": self.question, "answer": ""}},
                                       self.tokenizer, True) for c in cells]
        # one tokenizer call for all pages (the per-sample call of encode_item, batched: same ids and boxes per page)
        enc = self.tokenizer(text=[it[1] for it in items], text_pair=[it[2] for it in items],
                             boxes=[[list(map(float, b)) for b in it[3]] for it in items], padding=False, truncation=False)
        feats = [{"input_ids": torch.tensor(enc["input_ids"][k], dtype=torch.int64), "bbox": torch.tensor(enc["bbox"][k], dtype=torch.float32)}
                 for k in range(len(items))]
        batch = assembly.collate_for_generate(feats)
        return (texts, cells, batch["input_ids"].numpy().astype(np.int64), batch["bbox"].numpy().astype(np.float32),
                batch["attention_mask"].numpy().astype(np.int64))

    def _contexts(self, n):
        import threading
        from .inflight import InFlight
        if self._fl is None or len(self._fl) != n:
            if self._fl is not None:
                self._fl.close()
            self._emu_lock = threading.Lock()
            self._fl = InFlight(self.main, n, include_source=False, lock=self._emu_lock)
            for c in self._fl.contexts:
                c.set_stream_encoder(0)       # the contexts overlap one another: no run-ahead stream (and hardware queue) per context
        return self._fl

    def stage_main(self, pix, ids_in, bbox, mask, engine=None):
        main = engine if engine is not None else self.main
        B = int(ids_in.shape[0])
        mb = min(B, self.main_batch)
        if engine is None and self.main_inflight > 1 and B > mb:
            fl = self._contexts(self.main_inflight)
            # slabs of whole batches per context: the continuous decoder works through its slab, the batch form takes one batch per job
            per = mb * (-(-B // (mb * len(fl))) if (self.continuous and self.num_beams == 1) else 1)
            jobs = [(pix[c0:c0 + per], ids_in[c0:c0 + per], bbox[c0:c0 + per], mask[c0:c0 + per]) for c0 in range(0, B, per)]
            return self._stack_rows(fl.map(lambda ctx, a: self.stage_main(*a, engine=ctx), jobs))
        if self.continuous and self.num_beams == 1:
            out, lens, _ = main.generate_stream(ids_in, bbox, mask, pix, max_length=self.max_length, min_length=self.min_length,
                                                     chunk=mb, slots=mb, pool_chunks=3 if B > 2 * mb else 2)
            out = out.cpu().numpy() if hasattr(out, "cpu") else np.asarray(out)
            lens = lens.cpu().numpy() if hasattr(lens, "cpu") else np.asarray(lens)
            return out[:, :int(lens.max())]
        rows = []
        for c0 in range(0, B, mb):
            o, _, _ = main.generate(ids_in[c0:c0 + mb], bbox[c0:c0 + mb], mask[c0:c0 + mb], pix[c0:c0 + mb], num_beams=self.num_beams,
                                         max_length=self.max_length, min_length=self.min_length)
            rows.append(o.cpu().numpy() if hasattr(o, "cpu") else np.asarray(o))
        return self._stack_rows(rows)

    def _stack_rows(self, rows):
        width = max(r.shape[1] for r in rows)
        pad = self.main.shape.pad_token_id
        return np.concatenate([np.pad(r, ((0, 0), (0, width - r.shape[1])), constant_values=pad) for r in rows], axis=0)

    def __call__(self, pages_u8, timer: Optional[Callable[[], float]] = None) -> PipelineResult:
        t = {}
        now = timer or (lambda: 0.0)
        t0 = now()
        if self.overlap_slab > 0 and self.main_inflight > 1 and int(pages_u8.shape[0]) > self.overlap_slab:
            return self._call_overlapped(pages_u8, now)
        pix, new, steps = self.stage_ocr(pages_u8)
        if steps is not None:
            t["ocr_steps"] = steps
        t["ocr_s"] = now() - t0
        B, mb = int(new.shape[0]), self.main_batch
        if self.main_inflight > 1 and B > mb:
            # host stage and VTL stage pipelined: the pages go through the host stage a group at a time, the group's batches are handed to
            # the VTL contexts at once and decode while the next group is tokenised (continuous decoder: 4 batches' worth per context)
            import time as _time
            fl = self._contexts(self.main_inflight)
            per = mb * (4 if (self.continuous and self.num_beams == 1) else 1)
            group = per * len(fl)
            t1, host_busy, parts, futures = now(), 0.0, [], []
            for g0 in range(0, B, group):
                h0 = _time.perf_counter()
                part = self.stage_host(new[g0:g0 + group])
                host_busy += _time.perf_counter() - h0
                parts.append((new[g0:g0 + group],) + part)
                ids_in, bbox, mask = part[2:]
                for c0 in range(0, int(ids_in.shape[0]), per):
                    futures.append(fl.submit(lambda ctx, a: self.stage_main(*a, engine=ctx),
                                             (pix[g0 + c0:g0 + c0 + per], ids_in[c0:c0 + per], bbox[c0:c0 + per], mask[c0:c0 + per])))
            out = self._stack_rows([f.result() for f in futures])
            t["host_s"], t["main_s"] = host_busy, now() - t1
            return self._assemble(parts, out, t)
        t1 = now()
        texts, cells, ids_in, bbox, mask = self.stage_host(new)
        t["host_s"] = now() - t1
        t2 = now()
        out = self.stage_main(pix, ids_in, bbox, mask)
        t["main_s"] = now() - t2
        return PipelineResult(ids=out, ocr_new_ids=new, ocr_texts=texts, cells=cells, input_ids=ids_in, bbox=bbox, attention_mask=mask, timings=t)

    def _call_overlapped(self, pages_u8, now):
        """OCR of slab k + 1 (a worker thread driving the OCR contexts on their own streams) while the host stage and the VTL contexts work on
        slab k.  Same per-page results as the stage-after-stage loop: a page's OCR ids depend on the page only, its VTL ids on its own inputs."""
        import queue
        import threading
        import time as _time
        B, mb, slab = int(pages_u8.shape[0]), self.main_batch, self.overlap_slab
        fl = self._contexts(self.main_inflight)
        per = mb * (4 if (self.continuous and self.num_beams == 1) else 1)
        q = queue.Queue(maxsize=2)
        stat = {"ocr_busy": 0.0, "steps": 0}
        errs = []

        stop = threading.Event()

        def ocr_worker():
            try:
                for p0 in range(0, B, slab):
                    if stop.is_set():
                        break
                    h0 = _time.perf_counter()
                    pix, new, steps = self.stage_ocr(pages_u8[p0:p0 + slab], page0=p0)
                    stat["ocr_busy"] += _time.perf_counter() - h0
                    stat["steps"] = max(stat["steps"], steps or 0)
                    q.put((p0, pix, new))
            except BaseException as e:
                errs.append(e)
            finally:
                q.put(None)
        t0 = now()
        th = threading.Thread(target=ocr_worker, daemon=True)
        th.start()
        host_busy, parts, futures = 0.0, [], []
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                p0, pix, new = item
                h0 = _time.perf_counter()
                part = self.stage_host(new)
                host_busy += _time.perf_counter() - h0
                parts.append((new,) + part)
                ids_in, bbox, mask = part[2:]
                for c0 in range(0, int(ids_in.shape[0]), per):
                    futures.append(fl.submit(lambda ctx, a: self.stage_main(*a, engine=ctx),
                                             (pix[c0:c0 + per], ids_in[c0:c0 + per], bbox[c0:c0 + per], mask[c0:c0 + per])))
        except BaseException:
            # a failure of the host stage / a submit: stop the OCR worker (it may be blocked on the bounded queue) before re-raising
            stop.set()
            while th.is_alive():
                try:
                    q.get(timeout=0.1)
                except queue.Empty:
                    pass
            raise
        th.join()
        if errs:
            raise errs[0]
        out = self._stack_rows([f.result() for f in futures])
        t = {"ocr_s": stat["ocr_busy"], "host_s": host_busy, "main_s": now() - t0, "overlapped": True}
        if stat["steps"]:
            t["ocr_steps"] = stat["steps"]
        return self._assemble(parts, out, t)

    def _assemble(self, parts, out, timings):
        """parts: per group (ocr new ids, texts, cells, input_ids, bbox, attention_mask), each padded to its own widths."""
        width = max(p[3].shape[1] for p in parts)
        wnew = max(p[0].shape[1] for p in parts)
        padw = lambda a, v: np.pad(a, ((0, 0), (0, width - a.shape[1])) + ((0, 0),) * (a.ndim - 2), constant_values=v)
        return PipelineResult(
            ids=out, ocr_new_ids=np.concatenate([np.pad(p[0], ((0, 0), (0, wnew - p[0].shape[1])), constant_values=self.ocr.shape.pad_token_id)
                                                 for p in parts], axis=0),
            ocr_texts=[t for p in parts for t in p[1]], cells=[c for p in parts for c in p[2]],
            input_ids=np.concatenate([padw(p[3], self.main.shape.pad_token_id) for p in parts], axis=0),
            bbox=np.concatenate([padw(p[4], 0.0) for p in parts], axis=0), attention_mask=np.concatenate([padw(p[5], 0) for p in parts], axis=0),
            timings=timings)

    def close(self):
        for ctx, _ in self._ocr_ctx[1:]:
            ctx.close()
        self._ocr_ctx = []
        if self._fl is not None:
            self._fl.close()
            self._fl = None
        if self._prev_padding is not None:
            self.main.set_padding_semantics(self._prev_padding)
            self._prev_padding = None
