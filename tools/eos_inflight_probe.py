"""Probe: EOS-enabled queue (bench.py's eos run, EOS row x 12) through the continuous decoder on 4 contexts, 32 vs 64 slots per context.
    python tools/eos_inflight_probe.py"""
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
    sd = synth.recipe_state_dict(shape, **synth.BENCH_RECIPE)
    emb = sd["shared.weight"].copy()
    emb[shape.eos_token_id] = synth.round_bf16(emb[shape.eos_token_id] * np.float32(12.0))
    sd["shared.weight"] = emb
    eng = Engine(shape, max_decode_len=512)
    eng.load_state_dict(sd)
    B, QB = 32, 32
    inp = synth.synth_batch(shape, B, seed=synth.BENCH_SEED, return_pages=True)
    dev = {k: eng.mem.asarray(v, {"input_ids": np.int64, "bbox": np.float32, "attention_mask": np.uint8, "pages_u8": np.uint8}[k])
           for k, v in inp.items() if k != "pixel_values"}
    qd = {k: torch.cat([dev[k]] * QB, dim=0) for k in ("input_ids", "bbox", "attention_mask")}
    fl = InFlight(eng, 4)
    for c in fl.contexts:
        c.set_stream_encoder(0)
    per = QB * B // 4
    ref = None
    for slots in (32, 64, 32, 64):
        def job(ctx, i):
            sl = slice(i * per, (i + 1) * per)
            pix = torch.cat([ctx.preprocess(dev["pages_u8"]) for _ in range(per // B)], dim=0)
            o, l, st = ctx.generate_stream(qd["input_ids"][sl], qd["bbox"][sl], qd["attention_mask"][sl], pix, max_length=512, min_length=0,
                                           chunk=B, slots=slots, pool_chunks=3 if slots == 32 else 4)
            return o.cpu().numpy(), l.cpu().numpy(), st
        torch.cuda.synchronize(); t0 = time.time()
        res = fl.map(job, range(4))
        torch.cuda.synchronize(); dt = time.time() - t0
        ids = np.concatenate([r[0] for r in res]); lens = np.concatenate([r[1] for r in res])
        if ref is None:
            ref = (ids, lens)
        same = sum(int(lens[n] == ref[1][n] and np.array_equal(ids[n, :lens[n]], ref[0][n, :lens[n]])) for n in range(len(lens)))
        print(f"slots {slots}: {QB * B / dt:.1f} images/s, steps {[int(r[2]) for r in res]}, rows equal to the first run: {same}/{len(lens)}", flush=True)


if __name__ == "__main__":
    main()
