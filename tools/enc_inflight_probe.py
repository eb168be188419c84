"""Probe: the ENCODER alone with several contexts in flight.  One encoder call runs its GEMMs in whole rounds over the 256 CUs (320x256
tiles, one workgroup per CU) with every CU in its HBM-heavy epilogue or its MFMA-heavy main loop at the same time; a second context's
kernels can fill the tail rounds and put one kernel's epilogues beside another's main loops.  Prints ms per 32 images for
  * 1 context, batch 32 (the headline's encoder)     * k contexts, batch 32 each     * 2 contexts, batch 16 each (one batch cut in two)

    python tools/enc_inflight_probe.py [--reps 6]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    args = ap.parse_args()
    import torch
    from markushgrapher_amd import synth
    from markushgrapher_amd.engine import Engine
    from markushgrapher_amd.inflight import shared_streams

    shape = synth.SHAPES["large"]
    eng = Engine(shape, max_decode_len=64)
    eng.load_state_dict(synth.recipe_state_dict(shape, **synth.BENCH_RECIPE))
    engs = [eng] + [eng.clone() for _ in range(3)]
    inp = synth.synth_batch(shape, 32, seed=synth.BENCH_SEED, return_pages=True)
    pix = eng.preprocess(inp["pages_u8"])
    dt = {"input_ids": np.int64, "bbox": np.float32, "attention_mask": np.uint8}
    dev = {k: eng.mem.asarray(inp[k], dt[k]) for k in dt}
    streams = shared_streams(torch, eng.mem.device, 4)

    def worker(i, n, lo, hi):
        with torch.cuda.device(streams[i].device), torch.cuda.stream(streams[i]):
            for _ in range(n):
                engs[i].encode(dev["input_ids"][lo:hi], dev["bbox"][lo:hi], dev["attention_mask"][lo:hi], pix[lo:hi], want_out=False)
            streams[i].synchronize()

    def run(k, n, parts):
        torch.cuda.synchronize()
        t0 = time.time()
        th = [threading.Thread(target=worker, args=(i, n, *parts[i])) for i in range(k)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        return time.time() - t0

    full = [(0, 32)] * 4
    halves = [(0, 16), (16, 32)]
    quarters = [(0, 8), (8, 16), (16, 24), (24, 32)]
    for k, parts in ((4, full), (2, halves), (4, quarters)):
        run(k, 1, parts)          # warm-up
    t1 = run(1, args.reps, full)
    print("encoder, 1 context, batch 32: %.2f ms per 32 images" % (t1 / args.reps * 1e3), flush=True)
    for k in (2, 3, 4):
        tk = run(k, args.reps, full)
        print("encoder, %d contexts, batch 32 each: %.2f ms per 32 images" % (k, tk / (args.reps * k) * 1e3), flush=True)
    th = run(2, args.reps, halves)
    print("encoder, 2 contexts, batch 16 each (one batch in two halves): %.2f ms per 32 images" % (th / args.reps * 1e3), flush=True)
    tq = run(4, args.reps, quarters)
    print("encoder, 4 contexts, batch 8 each (one batch in four quarters): %.2f ms per 32 images" % (tq / args.reps * 1e3), flush=True)
    t16 = run(1, args.reps, halves)
    print("encoder, 1 context, batch 16: %.2f ms per 16 images" % (t16 / args.reps * 1e3), flush=True)


if __name__ == "__main__":
    main()
