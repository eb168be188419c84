"""GPU probe: continuous decoding (mg_generate_stream) against the batch loop, and where the run-ahead encoder should live.
    python tools/stream_probe.py [n_batches]
Prints images/s for: the batch loop (preprocess + mg_generate per 32 images), the stream with the encoder on the caller's stream
(mode 0), on its own low-priority stream (mode 1), and on CU-masked streams of several sizes / bit patterns (mode 2)."""
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch
from markushgrapher_amd import synth
from markushgrapher_amd.engine import Engine

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 5
new_tokens = int(sys.argv[2]) if len(sys.argv) > 2 else 256
shape = synth.SHAPES["large"]
sd = synth.recipe_state_dict(shape, **synth.BENCH_RECIPE)
eng = Engine(shape, max_decode_len=512)
eng.load_state_dict(sd)
B = 32
inp = synth.synth_batch(shape, B, seed=synth.BENCH_SEED, return_pages=True)
dev = {k: eng.mem.asarray(v, {"input_ids": np.int64, "bbox": np.float32, "attention_mask": np.uint8, "pages_u8": np.uint8}[k])
       for k, v in inp.items() if k != "pixel_values"}
N = nb * B
rep = lambda t: torch.cat([t] * nb, dim=0)
q = {k: rep(v) for k, v in dev.items()}
T = new_tokens + 1


def sync():
    torch.cuda.synchronize()


def batch_loop():
    for _ in range(nb):
        pix = eng.preprocess(dev["pages_u8"])
        ids, _, _ = eng.generate(dev["input_ids"], dev["bbox"], dev["attention_mask"], pix, max_length=T, min_length=T)
    return ids


def stream_run(min_len=T, max_len=T, slots=32, pool=3):
    pix = torch.cat([eng.preprocess(q["pages_u8"][c0:c0 + B]) for c0 in range(0, N, B)], dim=0)
    return eng.generate_stream(q["input_ids"], q["bbox"], q["attention_mask"], pix, max_length=max_len, min_length=min_len, chunk=B,
                               slots=slots, pool_chunks=pool)


def timed(fn, *a, **k):
    fn(*a, **k); sync()
    t0 = time.time(); r = fn(*a, **k); sync()
    return time.time() - t0, r


dt, ids_b = timed(batch_loop)
print(f"batch loop            : {N / dt:7.2f} images/s  ({dt / nb * 1e3:.1f} ms per 32 images)", flush=True)
ref = ids_b.cpu().numpy()
configs = [("stream mode 0 (serial)", 0, None), ("stream mode 1 (low-priority stream)", 1, None)]
for ncu, pat in (() if os.environ.get("QUICK") else ((32, "spread"), (64, "spread"), (96, "spread"), (128, "spread"), (64, "first"), (32, "first"))):
    bits = list(range(0, 256, 256 // ncu)) if pat == "spread" else list(range(ncu))
    configs.append((f"stream mode 2 ({ncu} CUs, {pat} bits)", 2, bits))
for name, mode, bits in configs:
    try:
        eng.set_stream_encoder(mode, cu_mask=bits)
        dt, (ids, lens, steps) = timed(stream_run)
        same = np.array_equal(ids.cpu().numpy()[:B], ref)
        print(f"{name:40s}: {N / dt:7.2f} images/s  ({dt / nb * 1e3:.1f} ms per 32 images, {steps} steps, ids equal batch: {same})", flush=True)
    except Exception as e:          # noqa: BLE001
        print(f"{name:40s}: FAILED {e}", flush=True)
if os.environ.get("QUICK"):
    sys.exit(0)
# EOS-enabled workload (rows end at different steps): batch generate vs the stream
emb = sd["shared.weight"].copy()
emb[shape.eos_token_id] = synth.round_bf16(sd["shared.weight"][shape.eos_token_id] * np.float32(12.0))
eng.load_state_dict({"shared.weight": emb})


def batch_eos():
    for _ in range(nb):
        pix = eng.preprocess(dev["pages_u8"])
        ids, _, _ = eng.generate(dev["input_ids"], dev["bbox"], dev["attention_mask"], pix, max_length=512)
    return ids


dt, ids_e = timed(batch_eos)
print(f"EOS run, batch loop   : {N / dt:7.2f} images/s (width {ids_e.shape[1]})", flush=True)
for mode in (0, 1):
    eng.set_stream_encoder(mode)
    dt, (ids, lens, steps) = timed(stream_run, 0, 512)
    ln = lens.cpu().numpy()
    print(f"EOS run, stream mode {mode}: {N / dt:7.2f} images/s ({steps} steps, mean length {ln.mean():.1f}, max {ln.max()})", flush=True)
