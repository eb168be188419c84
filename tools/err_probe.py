"""GPU-side probe: error metrics of the HIP path on the small golden fixtures (values behind the test tolerances).
MG_ROOT=<tree> python tools/err_probe.py"""
import os, sys
import numpy as np
ROOT = os.environ.get("MG_ROOT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
from markushgrapher_amd import synth
from markushgrapher_amd.engine import Engine
from oracle.udop_oracle import Oracle
G = os.path.join(ROOT, "tests", "golden")
for name in ("g0_tiny.npz", "g3_trained_tiny.npz", "g1_mid.npz"):
    g = dict(np.load(os.path.join(G, name)))
    shape = synth.SHAPES[str(g["shape"])]
    if name == "g3_trained_tiny.npz":
        sd = dict(np.load(os.path.join(G, "g3_weights.npz")))
    else:
        sd = synth.recipe_state_dict(shape, gain=float(g["gain"]))
    if "input_ids" in g:
        inp = {k: g[k] for k in ("input_ids", "bbox", "attention_mask", "pixel_values")}
    else:
        a = [int(v) for v in g["synth_args"]]
        inp = synth.synth_batch(shape, a[0], L_min=a[1], L_max=a[2], seed=a[3])
    eng = Engine(shape, max_decode_len=64)
    eng.load_state_dict(sd)
    enc, mask = eng.encode(inp["input_ids"], inp["bbox"], inp["attention_mask"], inp["pixel_values"])
    enc = eng.mem.numpy(enc)
    valid = g["enc_mask"].astype(bool)
    err = np.abs(enc - g["enc_out"])[valid]
    labels = g["labels"]
    dec_ids = Oracle.shift_right(labels, shape.decoder_start_token_id, shape.pad_token_id).numpy()
    dam = (labels != -100).astype(np.uint8)
    logits, _, _ = eng.forward_logits(inp["input_ids"], inp["bbox"], inp["attention_mask"], inp["pixel_values"], dec_ids, dam)
    lerr = np.abs(eng.mem.numpy(logits) - g["logits"])
    tol = 0.015 * float(np.abs(g["logits"]).max()) + 0.02
    print(f"{name}: enc err max {err.max():.4f} mean {err.mean():.5f} | logits err max {lerr.max():.4f} p99.9 {np.quantile(lerr, 0.999):.4f} mean {lerr.mean():.5f} tol {tol:.4f} max|logit| {np.abs(g['logits']).max():.2f}")
    if os.environ.get("MG_DUMP"):
        np.savez(os.path.join(os.environ["MG_DUMP"], name.replace(".npz", "") + "_dump.npz"), enc=enc, logits=eng.mem.numpy(logits))
