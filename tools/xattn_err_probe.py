"""Numerics of the two cross-attention forms (mg_set_cross_absorb) against the fp32 oracle on the trained tiny fixture, teacher-forced
along the golden ids for 511 positions (tests/test_engine.py::test_long_positions_forced_decode's set-up): max / mean |logit error| per form."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from markushgrapher_amd import synth  # noqa: E402
from oracle.udop_oracle import Oracle  # noqa: E402
from tests.backends import make_engine  # noqa: E402
from tests.conftest import load_golden  # noqa: E402
from tests.test_oracle_golden import _inputs, _weights  # noqa: E402

g = load_golden("g3_trained_tiny.npz")
shape, sd = _weights(g)
B, T = 6, 512
inp = {k: v[:B] for k, v in _inputs(g, shape).items()}
gi = g["greedy_ids"][:B]
forced = np.stack([np.resize(gi[b][gi[b] > 1], T) for b in range(B)])
forced[:, 0] = shape.decoder_start_token_id
ref = {}
for bf in (False, True):
    o = Oracle(shape, sd, emulate_bf16=bf)
    with torch.no_grad():
        enc, mask = o.encode(inp["input_ids"], inp["bbox"], inp["pixel_values"], inp["attention_mask"])
        hid, _ = o.decoder_stack(torch.from_numpy(forced[:, :T - 1]), mask, o.cross_kv(enc))
        ref[bf] = o.lm_logits(hid).numpy()
print("max |logit|", np.abs(ref[False]).max(), " fp32 vs bf16-emulating oracle: max", np.abs(ref[False] - ref[True]).max())
eng = make_engine("hip", shape, sd, max_decode_len=512)
for form in (1, 0):
    eng.set_cross_absorb(bool(form))
    cap = eng.debug_decode_capture(T - 1, B, forced)
    eng.generate(inp["input_ids"], inp["bbox"], inp["attention_mask"], inp["pixel_values"], max_length=T, min_length=T)
    c = eng.mem.numpy(cap).copy().transpose(1, 0, 2)
    eng.debug_decode_capture()
    err = np.abs(c - ref[False])
    per = err.max(axis=(0, 2))
    print(f"absorb={form}: max {err.max():.4f} (step {int(per.argmax())}), mean {err.mean():.5f}, steps<64 max {per[:64].max():.4f}, steps>=128 max {per[128:].max():.4f}, "
          f"vs bf16 oracle max {np.abs(c - ref[True]).max():.4f}")
