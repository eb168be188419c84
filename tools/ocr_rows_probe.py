"""Probe: the ChemicalOCR stage of the configs[4] loop alone (scripted lm_head: 512 pages of 10 - 120 cells, 683 new tokens on average, the
longest 1164), over (execution contexts, decode rows per context).  Prints pages/s, decode steps and decoded rows per ms.
    python tools/ocr_rows_probe.py [contexts:slots[:pages] ...]      e.g. 4:128 2:256 1:256 1:128:128
"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from markushgrapher_amd import synth
    from markushgrapher_amd.engine import Engine
    from markushgrapher_amd.ocr import OcrEngine
    from markushgrapher_amd.ocr_shapes import PRESETS, script_texts, scripted_state_dict, scripted_prompts, synth_cell_text, detokenize
    from markushgrapher_amd.pipeline import Configs4Pipeline
    from markushgrapher_amd.standin import make_udop_tokenizer
    pages_n, n_scripts = 512, 32
    eng = Engine(synth.SHAPES["large"], max_decode_len=64)          # (the pipeline object wants a VTL engine: the shared preprocessing)
    s = PRESETS["smoldocling"]
    n_cells = synth.randint("configs4/cells", n_scripts, 10, 120, synth.BENCH_SEED)
    texts = [synth_cell_text(int(n), synth.BENCH_SEED, f"p{i}") for i, n in enumerate(n_cells)]
    id_to_piece, chains, starts = script_texts(s, texts)
    ocr = OcrEngine(s).load_state_dict(scripted_state_dict(s, chains, starts))
    prompts = np.concatenate([scripted_prompts(s, chains, starts)] * (pages_n // n_scripts), axis=0)
    longest, total = max(len(c) for c in chains), sum(len(c) for c in chains) * (pages_n // n_scripts)
    pages = torch.from_numpy(synth.synth_pages_u8(32, 1024, synth.BENCH_SEED)).cuda()
    pages = torch.cat([pages] * (pages_n // 32), dim=0)
    cases = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(4, 128), (2, 256), (2, 192), (1, 256), (4, 64), (3, 171)]
    all_pages, all_prompts, all_total = pages, prompts, total
    for case in cases:
        ctxs, slots = case[:2]
        pages_n = case[2] if len(case) > 2 else 512
        pages, prompts, total = all_pages[:pages_n], all_prompts[:pages_n], all_total * pages_n // 512
        pipe = Configs4Pipeline(ocr, eng, make_udop_tokenizer(), lambda row: detokenize(id_to_piece, row, s.eos_token_id, s.pad_token_id), prompts,
                                ocr_max_new_tokens=longest + 8, ocr_slots=slots, ocr_inflight=ctxs, per_image_padding=False)
        pipe.stage_ocr(pages)
        torch.cuda.synchronize()
        t0 = time.time()
        pix, new, steps = pipe.stage_ocr(pages)
        torch.cuda.synchronize()
        dt = time.time() - t0
        ok = sum(detokenize(id_to_piece, new[i], s.eos_token_id, s.pad_token_id) == texts[i % n_scripts] for i in range(pages_n))
        print(f"contexts {ctxs} x {slots} rows: {pages_n / dt:7.1f} pages/s  {dt:6.3f} s  steps {steps}  {total / dt / 1e3:6.1f} new tokens per ms  "
              f"({steps and dt / steps * 1e3:.2f} ms per step and context)  strings as scripted {ok}/{pages_n}", flush=True)
        pipe.close()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
