"""Probe: the EOS-enabled queue (mg_generate_stream) over four contexts with 32 / 64 / 96 / 128 decode slots per context (bench.py's eos_enabled_continuous_in_flight
workload: 1024 images, EOS row scaled so that rows end at different steps); ids of every image compared with the 32-slot run.   python tools/eos_slots_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bench
    from markushgrapher_amd import synth
    from markushgrapher_amd.engine import Engine
    from markushgrapher_amd.inflight import InFlight
    shape = synth.SHAPES["large"]
    sd = synth.recipe_state_dict(shape, **synth.BENCH_RECIPE)
    eng = Engine(shape, max_decode_len=512)
    eng.load_state_dict(sd)
    B = 32
    inp = synth.synth_batch(shape, B, seed=synth.BENCH_SEED, return_pages=True)
    dev = {k: eng.mem.asarray(v, {"input_ids": np.int64, "bbox": np.float32, "attention_mask": np.uint8, "pixel_values": np.float32,
                                  "pages_u8": np.uint8}[k]) for k, v in inp.items()}
    emb = sd["shared.weight"].copy()
    scale = float(os.environ.get("EOS_SCALE", bench.EOS_ROW_SCALES[2] if len(bench.EOS_ROW_SCALES) > 2 else bench.EOS_ROW_SCALES[-1]))
    emb[shape.eos_token_id] = synth.round_bf16(sd["shared.weight"][shape.eos_token_id] * np.float32(scale))
    eng.load_state_dict({"shared.weight": emb})
    fl = InFlight(eng, 4)
    for c in fl.contexts:
        c.set_stream_encoder(0)
    QF = 32
    qf = {k: torch.cat([dev[k]] * QF, dim=0) for k in ("input_ids", "bbox", "attention_mask")}
    per = QF * B // len(fl)
    ref = None
    for slots in (32, 64, 96, 128):
        def job(ctx, i):
            sl = slice(i * per, (i + 1) * per)
            pix = torch.cat([ctx.preprocess(dev["pages_u8"]) for _ in range(per // B)], dim=0)
            o, l, st = ctx.generate_stream(qf["input_ids"][sl], qf["bbox"][sl], qf["attention_mask"][sl], pix, max_length=512, min_length=0,
                                           chunk=B, slots=slots, pool_chunks=2 + slots // B)
            return o.cpu().numpy(), l.cpu().numpy(), st
        fl.map(job, range(len(fl)))
        torch.cuda.synchronize(); t = time.time()
        res = fl.map(job, range(len(fl)))
        torch.cuda.synchronize(); t = time.time() - t
        ids = np.concatenate([r[0] for r in res]); lens = np.concatenate([r[1] for r in res])
        if ref is None:
            ref = (ids, lens)
        same = bool(np.array_equal(lens, ref[1]) and all(np.array_equal(ids[n, :lens[n]], ref[0][n, :lens[n]]) for n in range(len(lens))))
        print("slots %3d per context: %.1f images/s, steps per context %s, mean length %.1f, ids equal to the 32-slot run: %s" %
              (slots, QF * B / t, [int(r[2]) for r in res], float(lens.mean() - 1), same), flush=True)
        for c in fl.contexts:
            c.release_workspaces()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
