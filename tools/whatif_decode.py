"""What-if timing of the decode step in the headline regime (4 execution contexts x calls of 5 batches = 160 rows per decode step): the
TOOLS build of the library leaves launches of the step out by MG_WHATIF_STEP (engine.hip: 1 QKV, 2 self-attention, 4 [Wo | cross-Q],
8 cross-attention, 16 [Wxo | FFN-wi], 32 FFN-wo, 64 lm_head).  Results are WRONG, timing is valid: what the headline would be if a launch
cost nothing - the bound on what any faster form of that launch can give, alone and with the other contexts beside it.

    MG_WHATIF_STEP=32 python tools/whatif_decode.py [--inflight 4] [--batches-per-call 5] [--calls 1] [--new-tokens 256]
One mask per process (the switch is read once).  Prints images/s with all contexts busy and with one call alone.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inflight", type=int, default=4)
    ap.add_argument("--batches-per-call", type=int, default=5)
    ap.add_argument("--calls", type=int, default=1, help="timed calls per context")
    ap.add_argument("--new-tokens", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--product-lib", action="store_true", help="the product library (no what-if: the reference line)")
    ap.add_argument("--rows-split", type=int, default=-1, help="row-tile split policy of the decode projections (mgk_set_rows_split): -1 default, 0 never, 1 one row tile per workgroup")
    ap.add_argument("--resid-f16", type=int, default=1, help="0: residual projections with 8 instead of 16 features per workgroup (twice their activation bytes through L2)")
    args = ap.parse_args()
    import torch
    from markushgrapher_amd import synth
    from markushgrapher_amd.engine import Engine
    from markushgrapher_amd.inflight import InFlight

    shape = synth.SHAPES["large"]
    if args.product_lib:
        eng = Engine(shape, max_decode_len=512)
    else:
        from tools import _toolslib
        eng = Engine(shape, lib=_toolslib.load(), max_decode_len=512)
    eng.load_state_dict(synth.recipe_state_dict(shape, **synth.BENCH_RECIPE))
    if not args.resid_f16:
        eng.lib.mgk_set_resid_f16(0)
    if args.rows_split >= 0:
        eng.lib.mgk_set_rows_split(args.rows_split)
    B, nb, max_length = args.batch, args.batches_per_call, args.new_tokens + 1
    pool = [synth.synth_batch(shape, B, seed=synth.BENCH_SEED + 1000 * j, return_pages=True) for j in range(nb)]
    L = max(p["input_ids"].shape[1] for p in pool)
    for p in pool:
        n = L - p["input_ids"].shape[1]
        if n:
            p["input_ids"] = np.pad(p["input_ids"], ((0, 0), (0, n)))
            p["attention_mask"] = np.pad(p["attention_mask"], ((0, 0), (0, n)))
            p["bbox"] = np.pad(p["bbox"], ((0, 0), (0, n), (0, 0)))
    dt = {"input_ids": np.int64, "bbox": np.float32, "attention_mask": np.uint8, "pages_u8": np.uint8}
    src = {k: eng.mem.asarray(np.concatenate([p[k] for p in pool], axis=0), dt[k]) for k in dt}
    fl = InFlight(eng, args.inflight)

    def job(ctx):
        pix = ctx.preprocess(src["pages_u8"])
        out, _, _ = ctx.generate(src["input_ids"], src["bbox"], src["attention_mask"], pix, num_beams=1, max_length=max_length, min_length=max_length)
        return out

    def run(k, n):
        torch.cuda.synchronize()
        t0 = time.time()
        futs = [fl.submit(job) for _ in range(k * n)]
        for f in futs:
            f.result()
        torch.cuda.synchronize()
        return time.time() - t0

    run(len(fl), 1)
    t_all = run(len(fl), args.calls)
    t_one = run(1, 1)
    mask = int(os.environ.get("MG_WHATIF_STEP", "0")) if not args.product_lib else 0
    print(f"whatif {mask:3d}: {len(fl)} contexts x {args.calls} call(s) of {nb} batches: {B * nb * len(fl) * args.calls / t_all:7.1f} images/s;   one call alone: "
          f"{B * nb / t_one:7.1f} images/s ({t_one * 1e3:.0f} ms)", flush=True)
    fl.close()


if __name__ == "__main__":
    main()
