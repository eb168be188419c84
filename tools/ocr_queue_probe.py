"""GPU probe: ChemicalOCR queue form at the configs[4] shapes (scripted pages), for rocprofv3 kernel traces.
    python tools/ocr_queue_probe.py [pages] [slots]"""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from markushgrapher_amd import synth
from markushgrapher_amd.ocr import OcrEngine
from markushgrapher_amd.ocr_shapes import PRESETS, script_texts, scripted_state_dict, scripted_prompts, synth_cell_text, synth_inputs

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 128
s = PRESETS["smoldocling"]
n_cells = synth.randint("configs4/cells", 32, 10, 120, synth.BENCH_SEED)
texts = [synth_cell_text(int(n), synth.BENCH_SEED, f"p{i}") for i, n in enumerate(n_cells)]
v, chains, starts = script_texts(s, texts)
eng = OcrEngine(s).load_state_dict(scripted_state_dict(s, chains, starts))
prompts = np.concatenate([scripted_prompts(s, chains, starts)] * (N // 32), axis=0)
_, pix = synth_inputs(s, 32)
pix = torch.from_numpy(np.concatenate([pix] * (N // 32), axis=0)).cuda()
ids = torch.from_numpy(prompts).cuda()
longest = max(len(c) for c in chains)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    new, lens, steps = eng.generate_stream(ids, pix, longest + 8, slots=slots, chunk=128)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"{N} pages, {slots} slots: {N / dt:.1f} pages/s, {steps} steps, {dt / steps * 1e3:.3f} ms/step incl. prefill, mean len {lens.float().mean().item():.0f}", flush=True)
