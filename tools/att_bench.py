"""GPU microbenchmark of the encoder attention at the benchmark shape (B=32, H=16, S_cap=1280) through mg_encode's own call:
times `reps` launches with HIP events.  MG_ATT_DEPTH=1|2 selects the prefetch depth.  python tools/att_bench.py"""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
from markushgrapher_amd import synth
from markushgrapher_amd.engine import Engine
shape = synth.SHAPES["large"]
_tools = os.environ.get("MG_ATT_EXP") or os.environ.get("MG_GEMM_EXP") or os.environ.get("MG_TOOLS_LIB")
if os.environ.get("MG_LIB_PATH"):            # A/B against another build of the library (e.g. tools/_build/libmgrapher_prev.so)
    eng = Engine(shape, lib=C.CDLL(os.path.join(ROOT, os.environ["MG_LIB_PATH"])), max_decode_len=64)
elif _tools:                                 # what-if variants (WRONG results, timing only) live in the tools build
    from tools import _toolslib
    eng = Engine(shape, lib=_toolslib.load(), max_decode_len=64)
else:
    eng = Engine(shape, max_decode_len=64)
eng.load_state_dict(synth.recipe_state_dict(shape, **synth.BENCH_RECIPE))
inp = synth.synth_batch(shape, 32, seed=synth.BENCH_SEED, return_pages=True)
pix = eng.preprocess(inp["pages_u8"])
for _ in range(2):
    eng.encode(inp["input_ids"], inp["bbox"], inp["attention_mask"], pix, want_out=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    eng.encode(inp["input_ids"], inp["bbox"], inp["attention_mask"], pix, want_out=False)
e1.record(); torch.cuda.synchronize()
print("MG_ATT_EXP", os.environ.get("MG_ATT_EXP", "0"), "MG_GEMM_EXP", os.environ.get("MG_GEMM_EXP", "0"),
      "encoder ms per batch: %.2f" % (e0.elapsed_time(e1) / 5))
