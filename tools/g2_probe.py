import os, sys
import numpy as np
ROOT = os.environ.get("MG_ROOT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
from markushgrapher_amd import synth
from markushgrapher_amd.engine import Engine
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g2_large.npz")))
shape = synth.SHAPES["large"]
sd = synth.recipe_state_dict(shape, gain=float(g["gain"]))
inp = synth.synth_batch(shape, 1, seed=int(g["synth_seed"]), fixed_L=int(g["fixed_L"]))
eng = Engine(shape, max_decode_len=64)
eng.load_state_dict(sd)
enc, mask = eng.encode(inp["input_ids"], inp["bbox"], inp["attention_mask"], inp["pixel_values"])
enc, mask = eng.mem.numpy(enc)[0], eng.mem.numpy(mask)[0]
print("mask eq", np.array_equal(mask, g["enc_mask"][0].astype(np.uint8)), "valid", int(mask.sum()))
err = np.abs(enc[g["enc_rows"]] - g["enc_probe"])
print("rows", g["enc_rows"].tolist())
print("row max err", err.max(-1).round(4).tolist())
valid = mask.astype(bool)
print("abs sum rel", abs(np.abs(enc[valid]).astype(np.float64).sum() - float(g["enc_abs_sum"])) / float(g["enc_abs_sum"]), "finite", np.isfinite(enc).all())
if os.environ.get("MG_DUMP"):
    np.save(os.path.join(os.environ["MG_DUMP"], "g2_enc.npy"), enc)
