"""Probe: does a row's result depend on how many row tiles share its call?  The benchmark batch (32 images) decoded alone and as
rows 0-31 / 32-63 of ONE 64-row call (the same images twice): encoder output and greedy ids compared bit for bit; on a difference the
first differing step and the per-step top-2 margins are printed.   python tools/rows64_probe.py [--new-tokens 256] [--copies 2]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--new-tokens", type=int, default=256)
    ap.add_argument("--copies", type=int, default=2)
    args = ap.parse_args()
    import torch
    from markushgrapher_amd import synth
    from markushgrapher_amd.engine import Engine
    shape = synth.SHAPES["large"]
    eng = Engine(shape, max_decode_len=512)
    eng.load_state_dict(synth.recipe_state_dict(shape, **synth.BENCH_RECIPE))
    inp = synth.synth_batch(shape, 32, seed=synth.BENCH_SEED, return_pages=True)
    pix = eng.preprocess(inp["pages_u8"])
    L = args.new_tokens + 1
    enc1, _ = eng.encode(inp["input_ids"], inp["bbox"], inp["attention_mask"], pix)
    enc1 = enc1.cpu().numpy().copy()
    ids1, _, _ = eng.generate(inp["input_ids"], inp["bbox"], inp["attention_mask"], pix, max_length=L, min_length=L)
    ids1 = ids1.cpu().numpy().copy()
    c = args.copies
    rep = {k: np.concatenate([np.asarray(inp[k])] * c, 0) for k in ("input_ids", "bbox", "attention_mask")}
    pixc = torch.cat([pix] * c, 0)
    encc, _ = eng.encode(rep["input_ids"], rep["bbox"], rep["attention_mask"], pixc)
    encc = encc.cpu().numpy()
    for j in range(c):
        print("encoder output of copy %d equal to the 32-row call: %s" % (j, np.array_equal(encc[32 * j:32 * j + 32], enc1)), flush=True)
    idsc, _, _ = eng.generate(rep["input_ids"], rep["bbox"], rep["attention_mask"], pixc, max_length=L, min_length=L)
    idsc = idsc.cpu().numpy()
    for j in range(c):
        got = idsc[32 * j:32 * j + 32]
        bad = got != ids1
        print("ids of copy %d equal: %s  (%d rows differ, first differing step %d)" %
              (j, not bad.any(), int(bad.any(1).sum()), int(np.argmax(bad.any(0))) if bad.any() else -1), flush=True)


if __name__ == "__main__":
    main()
