"""Probe: one greedy call of --batches x 32 images alone on one context (weight-absorbed cross-attention pinned), a few decode steps, for a
rocprofv3 --kernel-trace run analysed by tools/rocpd_step_gaps.py (where a decode step's time goes: inside kernels / between them).

    python tools/alone_gaps_probe.py [--batches 5] [--new-tokens 24] [--contexts 1]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=5)
    ap.add_argument("--new-tokens", type=int, default=24)
    ap.add_argument("--contexts", type=int, default=1)
    ap.add_argument("--calls", type=int, default=2)
    args = ap.parse_args()
    import torch
    from markushgrapher_amd import synth
    from markushgrapher_amd.engine import Engine
    shape = synth.SHAPES["large"]
    sd = synth.recipe_state_dict(shape, **synth.BENCH_RECIPE)
    eng = Engine(shape, max_decode_len=512)
    eng.load_state_dict(sd)
    eng.set_cross_absorb(True)
    B = 32
    parts = [synth.synth_batch(shape, B, seed=synth.BENCH_SEED + 1000 * j, return_pages=True) for j in range(args.batches)]
    L = max(p["input_ids"].shape[1] for p in parts)

    def padto(x, L, v=0):
        if x.shape[1] == L:
            return x
        pad = [(0, 0), (0, L - x.shape[1])] + [(0, 0)] * (x.ndim - 2)
        return np.pad(x, pad, constant_values=v)
    inp = {k: np.concatenate([padto(p[k], L) if k in ("input_ids", "bbox", "attention_mask") else p[k] for p in parts], axis=0)
           for k in ("input_ids", "bbox", "attention_mask", "pages_u8")}
    dt = {"input_ids": np.int64, "bbox": np.float32, "attention_mask": np.uint8, "pages_u8": np.uint8}
    ml = args.new_tokens + 1
    if args.contexts > 1:
        from markushgrapher_amd.inflight import InFlight
        fl = InFlight(eng, args.contexts)
        ctxs = fl.contexts
    else:
        fl, ctxs = None, [eng]
    for c in ctxs:
        c.set_cross_absorb(True)
    dev = {k: eng.mem.asarray(inp[k], dt[k]) for k in dt}

    def call(ctx, i=0):
        pix = ctx.preprocess(dev["pages_u8"])
        ids, _, _ = ctx.generate(dev["input_ids"], dev["bbox"], dev["attention_mask"], pix, num_beams=1, max_length=ml, min_length=ml)
        return ids
    for _ in range(args.calls):
        torch.cuda.synchronize(); t0 = time.time()
        if fl is None:
            call(eng)
        else:
            fl.map(call, range(len(fl)))
        torch.cuda.synchronize()
        print("call of %d rows x %d contexts, %d new tokens: %.1f ms" % (B * args.batches, len(ctxs), args.new_tokens, (time.time() - t0) * 1e3), flush=True)


if __name__ == "__main__":
    main()
