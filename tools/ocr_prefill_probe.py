"""ChemicalOCR stage, vision tower + prefill alone: 32 pages (SmolDocling-256M geometry, recipe weights), generate() with ONE new token, 20 times.
    python tools/ocr_prefill_probe.py [B]"""
import dataclasses
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

if __name__ == "__main__":
    import torch
    from markushgrapher_amd.ocr import OcrEngine
    from markushgrapher_amd.ocr_shapes import PRESETS, recipe_state_dict, synth_inputs
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    s = dataclasses.replace(PRESETS["smoldocling"], eos_token_id=-1)
    eng = OcrEngine(s).load_state_dict(recipe_state_dict(s))
    ids, pix = synth_inputs(s, B)
    ids, pix = torch.from_numpy(ids).cuda(), torch.from_numpy(pix).cuda()
    for _ in range(3):
        eng.generate(ids, pix, 1)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20):
        eng.generate(ids, pix, 1)
    torch.cuda.synchronize()
    print(f"vision tower + prefill + first token, {B} pages: {(time.time() - t0) / 20 * 1e3:.2f} ms per call")
