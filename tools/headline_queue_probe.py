"""Probe: the headline workload (batches of 32 different images, 256 forced new tokens) through the CONTINUOUS decoder (mg_generate_stream) on 4
execution contexts: each context takes 160 images as chunks of 32 into 160 decode slots, the encoder of the later chunks runs ahead on the
context's second stream while the rows of the earlier chunks already decode (encoder: MFMA-bound, decode: HBM-bound).  Against the batch
form (mg_generate on calls of 5 batches: encoder, then decode).
    python tools/headline_queue_probe.py"""
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    import torch
    from markushgrapher_amd import synth
    from markushgrapher_amd.engine import Engine
    from markushgrapher_amd.inflight import InFlight
    shape = synth.SHAPES["large"]
    eng = Engine(shape, max_decode_len=512)
    eng.load_state_dict(synth.recipe_state_dict(shape, **synth.BENCH_RECIPE))
    B, NB, NEW = 32, 5, 256
    pool = [synth.synth_batch(shape, B, seed=synth.BENCH_SEED + 1000 * j, return_pages=True) for j in range(4 * NB)]
    L = max(p["input_ids"].shape[1] for p in pool)
    for p in pool:
        n = L - p["input_ids"].shape[1]
        if n:
            p["input_ids"] = np.pad(p["input_ids"], ((0, 0), (0, n)))
            p["attention_mask"] = np.pad(p["attention_mask"], ((0, 0), (0, n)))
            p["bbox"] = np.pad(p["bbox"], ((0, 0), (0, n), (0, 0)))
    dt = {"input_ids": np.int64, "bbox": np.float32, "attention_mask": np.uint8, "pages_u8": np.uint8}
    calls = [{k: eng.mem.asarray(np.concatenate([p[k] for p in pool[i * NB:(i + 1) * NB]], axis=0), dt[k]) for k in dt} for i in range(4)]
    fl = InFlight(eng, 4)
    for c in fl.contexts:
        c.set_cross_absorb(True)

    def job_batch(ctx, i):
        src = calls[i]
        pix = ctx.preprocess(src["pages_u8"])
        return ctx.generate(src["input_ids"], src["bbox"], src["attention_mask"], pix, max_length=NEW + 1, min_length=NEW + 1)[0].cpu().numpy()

    def job_queue(ctx, i, chunk, slots, mode):
        ctx.set_stream_encoder(mode)
        src = calls[i]
        pix = ctx.preprocess(src["pages_u8"])
        o, l, st = ctx.generate_stream(src["input_ids"], src["bbox"], src["attention_mask"], pix, max_length=NEW + 1, min_length=NEW + 1,
                                       chunk=chunk, slots=slots, pool_chunks=max(2, (slots + chunk - 1) // chunk))
        return o.cpu().numpy(), int(st)

    ref = fl.map(job_batch, range(4))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        ref = fl.map(job_batch, range(4))
        torch.cuda.synchronize(); t = time.time() - t0
        print(f"batch form: {4 * NB * B / t:.1f} images/s", flush=True)
    for chunk, slots, mode in ((32, 160, 1), (32, 160, 0), (64, 160, 1), (32, 128, 1), (16, 160, 1)):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.time()
            res = fl.map(lambda ctx, i: job_queue(ctx, i, chunk, slots, mode), range(4))
            torch.cuda.synchronize(); t = time.time() - t0
            same = sum(int(np.array_equal(res[i][0][:, :NEW + 1], ref[i])) for i in range(4))
            print(f"queue form chunk {chunk} slots {slots} encoder mode {mode}: {4 * NB * B / t:.1f} images/s, steps {[r[1] for r in res]}, calls equal to the batch form {same}/4", flush=True)


if __name__ == "__main__":
    main()
