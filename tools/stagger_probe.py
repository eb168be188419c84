"""Probe: the headline plan (4 contexts x one call of 5 batches, 256 forced tokens) with the contexts' calls submitted at different times:
the first `k` contexts at t = 0, the others `delay` ms later, so that their encoder phases (MFMA-bound) run beside the first ones' decode
steps (HBM-bound) instead of beside each other.  Time = first submission -> all calls done (20 batches).
    python tools/stagger_probe.py"""
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

    def job(ctx, i):
        src = calls[i]
        pix = ctx.preprocess(src["pages_u8"])
        return ctx.generate(src["input_ids"], src["bbox"], src["attention_mask"], pix, max_length=NEW + 1, min_length=NEW + 1)[0].cpu().numpy()

    ref = fl.map(job, range(4))
    for first, delay in ((4, 0), (2, 0.2), (2, 0.35), (2, 0.5), (1, 0.17), (3, 0.5), (4, 0), (2, 0.35)):
        torch.cuda.synchronize(); t0 = time.time()
        futs = [fl.submit(job, i) for i in range(first)]
        rest = list(range(first, 4))
        if first == 1:          # one by one
            for i in rest:
                time.sleep(delay)
                futs.append(fl.submit(job, i))
        elif rest:
            time.sleep(delay)
            futs += [fl.submit(job, i) for i in rest]
        res = [f.result() for f in futs]
        torch.cuda.synchronize(); t = time.time() - t0
        same = sum(int(np.array_equal(res[i], ref[i])) for i in range(4))
        print(f"first {first} contexts at 0, the others {'each ' if first == 1 else ''}{delay * 1e3:.0f} ms later: {4 * NB * B / t:.1f} images/s ({t * 1e3:.0f} ms), ids equal {same}/4", flush=True)


if __name__ == "__main__":
    main()
