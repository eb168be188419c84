"""GPU probe: main-model continuous decoder with more decode slots than the encoder chunk (forced 256 tokens, 8 x 32 images)."""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from markushgrapher_amd import synth
from markushgrapher_amd.engine import Engine
shape = synth.SHAPES["large"]
eng = Engine(shape, max_decode_len=512)
eng.load_state_dict(synth.recipe_state_dict(shape, **synth.BENCH_RECIPE))
B, nb, T = 32, 8, 257
inp = synth.synth_batch(shape, B, seed=synth.BENCH_SEED, return_pages=True)
dev = {k: eng.mem.asarray(v, {"input_ids": np.int64, "bbox": np.float32, "attention_mask": np.uint8, "pages_u8": np.uint8}[k]) for k, v in inp.items() if k != "pixel_values"}
q = {k: torch.cat([v] * nb, dim=0) for k, v in dev.items()}
pix = torch.cat([eng.preprocess(dev["pages_u8"])] * nb, dim=0)
ref = None
for slots in (32, 64, 96, 128):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        ids, lens, steps = eng.generate_stream(q["input_ids"], q["bbox"], q["attention_mask"], pix, max_length=T, min_length=T, chunk=B, slots=slots, pool_chunks=max(3, slots // B + 2))
        torch.cuda.synchronize(); dt = time.time() - t0
    a = ids.cpu().numpy()
    if ref is None: ref = a
    print(f"slots {slots:3d}: {nb * B / dt:6.1f} images/s, {steps} steps, {dt / steps * 1e3:.3f} ms/step; rows equal to the 32-slot run: {int((a == ref).all(1).sum())}/{nb * B}", flush=True)
