"""Probe: N independent mg_generate calls in flight (one model handle, host thread, stream and workspace each) versus the same
number of batches run back to back on one handle.  The decode step is launch/latency-bound for 5 of its 6 launches per layer,
so a second batch's launches can fill the idle machine; rows are independent, so ids are identical by construction.

    python tools/inflight_probe.py [--inflight 2] [--batches 4] [--new-tokens 256]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inflight", type=int, default=2)
    ap.add_argument("--batches", type=int, default=4, help="batches per handle")
    ap.add_argument("--new-tokens", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--clones", action="store_true", help="contexts = clones of one engine (shared weights) instead of separately loaded engines")
    ap.add_argument("--stagger-ms", type=float, default=0.0, help="start context i that many ms x i late")
    ap.add_argument("--tools-lib", action="store_true", help="the tools build of the library (what-if variants: MG_WHATIF_KV ...)")
    ap.add_argument("--only", action="store_true", help="measure the full in-flight count only (plus the one-at-a-time reference)")
    args = ap.parse_args()
    import torch
    from markushgrapher_amd import synth
    from markushgrapher_amd.engine import Engine

    shape = synth.SHAPES["large"]
    sd = synth.recipe_state_dict(shape, **synth.BENCH_RECIPE)
    B, max_length = args.batch, args.new_tokens + 1
    engs = []
    for i in range(args.inflight):
        if args.clones and engs:
            engs.append(engs[0].clone())
            continue
        if args.tools_lib:
            from tools import _toolslib
            e = Engine(shape, lib=_toolslib.load(), max_decode_len=512)
        else:
            e = Engine(shape, max_decode_len=512)
        e.load_state_dict(sd)
        engs.append(e)
    inp = synth.synth_batch(shape, B, seed=synth.BENCH_SEED, return_pages=True)
    dt = {"input_ids": np.int64, "bbox": np.float32, "attention_mask": np.uint8, "pixel_values": np.float32, "pages_u8": np.uint8}
    dev = {k: engs[0].mem.asarray(v, dt[k]) for k, v in inp.items()}
    streams = [torch.cuda.Stream() for _ in engs]
    results = [None] * len(engs)

    def worker(i, n):
        if args.stagger_ms > 0:
            time.sleep(i * args.stagger_ms * 1e-3)
        with torch.cuda.stream(streams[i]):
            for _ in range(n):
                pix = engs[i].preprocess(dev["pages_u8"])
                ids, _, _ = engs[i].generate(dev["input_ids"], dev["bbox"], dev["attention_mask"], pix, num_beams=1,
                                             max_length=max_length, min_length=max_length)
            results[i] = ids

    def run(k, n):
        torch.cuda.synchronize()
        t0 = time.time()
        th = [threading.Thread(target=worker, args=(i, n)) for i in range(k)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        return time.time() - t0

    run(len(engs), 1)          # warm-up (graph capture on every handle)
    t1 = run(1, args.batches)
    print("1 in flight: %.1f ms/batch  %.2f images/s" % (t1 / args.batches * 1e3, B * args.batches / t1), flush=True)
    ref = results[0].cpu().numpy()
    print("   one at a time, rows 0-1, first 8 ids:", ref[:2, :8].tolist(), flush=True)
    for k in range(len(engs) if args.only else 2, len(engs) + 1):
        tk = run(k, args.batches)
        same = all(np.array_equal(results[i].cpu().numpy(), ref) for i in range(k))
        if not same:
            for i in range(k):
                r = results[i].cpu().numpy()
                print("   context %d rows 0-1:" % i, r[:2, :8].tolist(), flush=True)
                bad = r != ref
                first = int(np.argmax(bad.any(axis=0))) if bad.any() else -1
                print("   context %d: %d of %d rows differ, first differing step %d" % (i, int(bad.any(axis=1).sum()), r.shape[0], first), flush=True)
        print("%d in flight: %.1f ms/batch  %.2f images/s  (x%.2f)  ids equal: %s" %
              (k, tk / (k * args.batches) * 1e3, B * k * args.batches / tk, t1 * k / tk, same), flush=True)


if __name__ == "__main__":
    main()
