"""Probe: the reference's default decode mode (beam-5, EOS live, max_length 512) as batch calls against the beam queue
(mg_generate_stream_beam), one context.   python tools/beam_queue_probe.py [--slots 32] [--queue 8]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, nargs="+", default=[32, 48])
    ap.add_argument("--queue", type=int, default=8, help="batches of 32 images in the queue")
    ap.add_argument("--eos-scale", type=float, default=12.0)
    ap.add_argument("--contexts", type=int, default=1, help="> 1: the queue on that many execution contexts at once (markushgrapher_amd/inflight.py), a queue of --queue batches each")
    ap.add_argument("--tools-lib", action="store_true", help="the tools build of the library (MG_WHATIF_STEP what-if runs: wrong results, valid timing)")
    args = ap.parse_args()
    import torch
    from markushgrapher_amd import synth
    from markushgrapher_amd.engine import Engine
    shape = synth.SHAPES["large"]
    sd = synth.recipe_state_dict(shape, **synth.BENCH_RECIPE)
    emb = sd["shared.weight"].copy()
    emb[shape.eos_token_id] = synth.round_bf16(sd["shared.weight"][shape.eos_token_id] * np.float32(args.eos_scale))
    sd["shared.weight"] = emb
    if args.tools_lib:
        from tools import _toolslib
        eng = Engine(shape, lib=_toolslib.load(), max_decode_len=512)
    else:
        eng = Engine(shape, max_decode_len=512)
    eng.load_state_dict(sd)
    B = 32
    inp = synth.synth_batch(shape, B, seed=synth.BENCH_SEED, return_pages=True)
    dt = {"input_ids": np.int64, "bbox": np.float32, "attention_mask": np.uint8, "pages_u8": np.uint8}
    dev = {k: eng.mem.asarray(inp[k], dt[k]) for k in dt}

    def batch():
        out, sc, _ = eng.generate(dev["input_ids"], dev["bbox"], dev["attention_mask"], eng.preprocess(dev["pages_u8"]), num_beams=5,
                                  max_length=512, min_length=0)
        return out.cpu().numpy(), sc.cpu().numpy()
    batch()
    torch.cuda.synchronize(); t0 = time.time()
    ref, ref_sc = batch()
    torch.cuda.synchronize(); tb = time.time() - t0
    print("batch call: %.1f ms per 32 images = %.2f images/s, %d columns" % (tb * 1e3, B / tb, ref.shape[1]), flush=True)
    Q = args.queue
    q = {k: torch.cat([dev[k]] * Q, dim=0) for k in ("input_ids", "bbox", "attention_mask")}
    eng.set_stream_encoder(0)
    if args.contexts > 1:
        from markushgrapher_amd.inflight import InFlight
        fl = InFlight(eng, args.contexts)
        for c in fl.contexts:
            c.set_stream_encoder(0)
        for slots in args.slots:
            def job(ctx, i):
                pix = torch.cat([ctx.preprocess(dev["pages_u8"]) for _ in range(Q)], dim=0)
                o, l, sc, st = ctx.generate_stream_beam(q["input_ids"], q["bbox"], q["attention_mask"], pix, num_beams=5, max_length=512, min_length=0,
                                                        chunk=B, slots=slots, pool_chunks=3)
                return o.cpu().numpy(), l.cpu().numpy(), st
            fl.map(job, range(len(fl)))
            torch.cuda.synchronize(); t0 = time.time()
            res = fl.map(job, range(len(fl)))
            torch.cuda.synchronize(); tq = time.time() - t0
            same = all(np.array_equal(o[n, :min(int(l[n]), ref.shape[1])], ref[n % B, :min(int(l[n]), ref.shape[1])]) for o, l, _ in res for n in range(Q * B))
            print("queue on %d contexts, %d image slots each: %.2f images/s, steps %s, hypotheses equal %s"
                  % (len(fl), slots, len(fl) * Q * B / tq, [int(r[2]) for r in res], same), flush=True)
        fl.close()
        return
    for slots in args.slots:
        def queue():
            pix = torch.cat([eng.preprocess(dev["pages_u8"]) for _ in range(Q)], dim=0)
            return eng.generate_stream_beam(q["input_ids"], q["bbox"], q["attention_mask"], pix, num_beams=5, max_length=512, min_length=0,
                                            chunk=B, slots=slots, pool_chunks=3)
        queue()
        torch.cuda.synchronize(); t0 = time.time()
        o, l, sc, steps = queue()
        torch.cuda.synchronize(); tq = time.time() - t0
        o, l, sc = o.cpu().numpy(), l.cpu().numpy(), sc.cpu().numpy()
        same = all(np.array_equal(o[n, :min(int(l[n]), ref.shape[1])], ref[n % B, :min(int(l[n]), ref.shape[1])]) for n in range(Q * B))
        same_sc = bool(np.array_equal(sc, np.tile(ref_sc, Q)))
        print("queue, %d image slots: %.1f ms per 32 images = %.2f images/s (%.2f x), %d steps, mean length %.1f, hypotheses equal %s, scores equal %s"
              % (slots, tq / Q * 1e3, Q * B / tq, (Q * B / tq) / (B / tb), steps, l.mean(), same, same_sc), flush=True)


if __name__ == "__main__":
    main()
