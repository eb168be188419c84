"""GPU-side probe: is the encoder bitwise reproducible run to run?  (races show up as differing bits)"""
import os, sys
import numpy as np
ROOT = os.environ.get("MG_ROOT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
from markushgrapher_amd import synth
from markushgrapher_amd.engine import Engine
shape = synth.SHAPES["large"]
sd = synth.recipe_state_dict(shape, **synth.BENCH_RECIPE)
eng = Engine(shape, max_decode_len=64)
eng.load_state_dict(sd)
inp = synth.synth_batch(shape, 32, seed=synth.BENCH_SEED, return_pages=True)
pix = eng.preprocess(inp["pages_u8"])
outs = []
for i in range(4):
    enc, mask = eng.encode(inp["input_ids"], inp["bbox"], inp["attention_mask"], pix)
    outs.append(eng.mem.numpy(enc).copy())
m = eng.mem.numpy(mask).astype(bool)
for i in range(1, 4):
    d = np.abs(outs[i] - outs[0])
    print(f"run {i} vs 0: differing elements {int((d > 0).sum())} max diff {d.max():.4g}; valid rows only: {int((d[m] > 0).sum())} max {d[m].max():.4g}")
for i in range(1, 4):
    for j in range(i + 1, 4):
        d = np.abs(outs[i] - outs[j])
        print(f"run {i} vs {j}: differing {int((d > 0).sum())} max {d.max():.4g}")
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g4_bench.npz")))
for i in range(4):
    pe = np.stack([np.abs(outs[i][b][g["enc_rows"][b]] - g["enc_probe"][b]) for b in range(32)])
    print(f"run {i}: probe rows vs stock max {pe.max():.4f} mean {pe.mean():.5f}")
