"""Feasibility probe: does running two independent 16-image decodes concurrently on two HIP streams beat one 32-image
decode?  (The decode step is a chain of latency-bound launches; two chains can interleave on the 256 CUs.)"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from markushgrapher_amd import synth  # noqa: E402
from markushgrapher_amd.engine import Engine  # noqa: E402


def main():
    shape = synth.SHAPES["large"]
    sd = synth.recipe_state_dict(shape, gain=1.0)
    T = 257
    inp = synth.synth_batch(shape, 32, seed=20260928)
    engs = [Engine(shape, max_decode_len=512) for _ in range(2)]
    for e in engs:
        e.load_state_dict(sd)
    dt = {"input_ids": np.int64, "bbox": np.float32, "attention_mask": np.uint8, "pixel_values": np.float32}

    def dev(e, sl):
        return {k: e.mem.asarray(v[sl], dt[k]) for k, v in inp.items() if k in dt}

    full = dev(engs[0], slice(0, 32))
    halves = [dev(engs[i], slice(16 * i, 16 * i + 16)) for i in range(2)]

    def gen(e, d):
        return e.generate(d["input_ids"], d["bbox"], d["attention_mask"], d["pixel_values"], num_beams=1, max_length=T, min_length=T)[0]

    def timed(fn, n=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.time() - t0) / n * 1e3

    ref = gen(engs[0], full)
    print("one stream, B=32:        %.1f ms" % timed(lambda: gen(engs[0], full)))
    print("one stream, 2 x B=16:    %.1f ms" % timed(lambda: (gen(engs[0], halves[0]), gen(engs[1], halves[1]))))
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = [None, None]

    def worker(i):
        with torch.cuda.stream(streams[i]):
            outs[i] = gen(engs[i], halves[i])

    def both():
        th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    print("two streams, 2 x B=16:   %.1f ms" % timed(both))
    for e in engs:
        e.set_decode_graph(0)
    print("eager: one stream 2 x B=16: %.1f ms" % timed(lambda: (gen(engs[0], halves[0]), gen(engs[1], halves[1]))))
    print("eager: two streams:         %.1f ms" % timed(both))
    got = torch.cat([outs[0], outs[1]], 0)
    print("ids identical to B=32 run:", bool(torch.equal(got.cpu(), ref.cpu())))


if __name__ == "__main__":
    main()
