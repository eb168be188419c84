#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): images/s, greedy CXSMILES decode, 1024 px crops, batch 32 per GPU.

One "step" = one pass of the hot path over one batch of 32 synthetic pages per GPU: device LANCZOS resize of the
1024 px u8 crops to the model's 512 px input (mg_preprocess_pages), VTL encoder, cross-K/V projection and 256 greedy
decode steps (EOS suppressed: min_length = max_length = 257) on the UDOP-large-shaped
MarkushGrapher-2 model with the recipe weights of tests/golden/g4_bench.npz (synth.BENCH_RECIPE: the configuration the
parity tests pin on stock UDOP).  Inputs are resident in HBM when the timed region starts.  With --gpus N every rank
runs its own 32-image shard (weak scaling) and the decoded ids are all-gathered over RCCL inside the timed region.
Every rank keeps --inflight (default 4) execution contexts going at once (mg_clone: same weights, own workspace / decode graph /
stream / host thread), and a call of a context takes up to --batches-per-call (default 5) batches of 32 side by side (the decode
step then reads the decoder's weights once for all of them): the K timed steps are cut into near-equal calls over the contexts
(inflight.plan_calls: 20 steps = one call of 5 batches per context), all of them start and end inside the timed region, every
batch's ids are bit-identical to a call on the batch alone (tests/test_engine.py, tests/test_inflight.py; checked in every run:
config.ids_equal_one_batch_calls).  `one_call_alone` repeats the largest call shape on one context, `one_batch_in_flight` the
one-batch-at-a-time loop of the reference.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0:
  roofline       dominant kernel (single-query cross-attention over the per-image K/V stream), algorithmic bytes / average
                 launch duration from HIP events on the launch stream of the first context during the timed region (i.e. beside
                 the other batches' kernels); `traffic` = HBM bytes per launch from a FETCH_SIZE pass (rocprofv3 --pmc, run by this
                 script as a child on a short copy of the workload when rocprofv3 is present)
  one_call_alone        one call of the timed region's largest shape alone on one context: the dominant launch and the phases uncontended
  one_batch_in_flight   one batch per call on one context: images/s, the kernel's and the phases' figures (the loop of rounds 1-2)
  phases         encoder (MFMA-bound) and decode step (HBM-bound) against their own rooflines: enc_mfma_frac, dec_hbm_frac,
                 dec_mfma_frac (SURVEY.md §8d formulas), phase times from HIP events inside mg_generate
  extra_runs     EOS-enabled greedy run (max_length 512) and beam-5 (BASELINE configs[2]) on the same inputs
  cpu_baseline   the fp32 CPU oracle (oracle/udop_oracle.py, kind "port") on a bounded sample on this box's host cores
"""
import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

# before torch loads the HIP runtime: kernel arguments in device memory (see markushgrapher_amd/__init__.py)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# one hardware queue per execution context of the batches in flight (same file)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "images/sec whole-node (greedy CXSMILES decode, 1024px crops, bs=32/GPU)"
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (same guide)
SOLO_STEPS = 3                                      # batches of the one-batch-in-flight side measurement
EOS_ROW_SCALES = (3.0, 4.0, 6.0, 8.0, 12.0, 16.0)   # EOS-enabled run: ladder of scales of the EOS embedding row (see extra_runs)


def cpu_baseline(shape, sd, L=128, new_tokens=128, sample_steps=8, reps=3):
    """fp32 CPU oracle (kind "port"), SURVEY.md §8d protocol bounded to ~30 s: B = 1 and B = 4, L_text = 128, greedy;
    encoder + cross-K/V once per batch size, then 1 warm-up + `reps` timed repetitions of `sample_steps` decode steps,
    per-step time extrapolated to 128 new tokens (a step's cost does not depend on the position: the cross-attention
    over ~1150 encoder positions dominates the growing self-attention cache)."""
    import torch
    from markushgrapher_amd import synth
    from oracle.udop_oracle import Oracle
    # a small batch on a 100+-core host is slower with every core than with a few dozen threads (memory-bound GEMV)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    o = Oracle(shape, sd)
    out = {}
    for B in (1, 4):
        inp = synth.synth_batch(shape, B, seed=7, fixed_L=L)
        with torch.no_grad():
            t0 = time.time()
            enc, mask = o.encode(inp["input_ids"], inp["bbox"], inp["pixel_values"], inp["attention_mask"])
            xkv = o.cross_kv(enc)
            t_enc = time.time() - t0

            def run(n):
                kv, cur = None, torch.zeros((B, 1), dtype=torch.long)
                t1 = time.time()
                for t in range(n):
                    hid, kv = o.decoder_stack(cur, mask, xkv, kv, t)
                    cur = torch.argmax(o.lm_logits(hid[:, -1:, :])[:, 0, :], dim=-1)[:, None]
                return (time.time() - t1) / n
            run(1)
            t_step = min(run(sample_steps) for _ in range(reps))
        out[B] = (B / (t_enc + new_tokens * t_step), t_enc, t_step)
    return {"value": round(out[4][0], 5), "unit": "images/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "value_b1": round(out[1][0], 5),
            "sample": f"oracle fp32 torch-CPU ({torch.__version__}), UDOP-large shape, L_text={L}, greedy, {new_tokens} new tokens: "
                      f"B=1 encoder+cross-KV {out[1][1]:.2f}s + {out[1][2] * 1e3:.1f} ms/step; B=4 {out[4][1]:.2f}s + "
                      f"{out[4][2] * 1e3:.1f} ms/step (1 warm-up + {reps} x {sample_steps} timed decode steps, best rep, "
                      f"extrapolated to {new_tokens} steps); value = B=4"}


def ocr_stage_run(B=32, new_tokens=256):
    """SURVEY.md §8 row f-1 (BASELINE configs[4] names the stage): ChemicalOCR = an Idefics3-class VLM, SmolDocling-256M geometry
    (INFERRED), one 512-px page per sequence, greedy.  EOS cannot occur (eos id -1), so the work is fixed: vision tower +
    prompt prefill + `new_tokens` KV-cached steps.  Vision tower and prefill are in their second form (tiled fp32 residual streams, residual projections on the batched epilogue, tile-wise LayerNorm / RMSNorm,
    the tower's attention on the encoder's second-form kernel without its bias path); the decode step runs on the main path's
    deferred-RMSNorm kernels, 4 launches per layer, replayed as a HIP graph."""
    import dataclasses
    import torch
    from markushgrapher_amd.ocr import OcrEngine
    from markushgrapher_amd.ocr_shapes import PRESETS, recipe_state_dict, synth_inputs
    s = dataclasses.replace(PRESETS["smoldocling"], eos_token_id=-1)
    eng = OcrEngine(s).load_state_dict(recipe_state_dict(s))
    ids, pix = synth_inputs(s, B)
    ids, pix = torch.from_numpy(ids).cuda(), torch.from_numpy(pix).cuda()

    def run(n):
        torch.cuda.synchronize(); t0 = time.time()
        eng.generate(ids, pix, n)
        torch.cuda.synchronize()
        return time.time() - t0
    run(2)
    t1 = min(run(1), run(1))                  # vision tower + prefill + first selection
    tn = min(run(new_tokens), run(new_tokens))
    step_ms = (tn - t1) / (new_tokens - 1) * 1e3
    wbytes = 2 * (s.t_layers * ((s.t_heads + 2 * s.t_kv_heads) * 64 * s.t_hidden + s.t_hidden * s.t_hidden + 3 * s.t_inter * s.t_hidden) + s.vocab * s.t_hidden)
    L = int(ids.shape[1])
    kvbytes = B * s.t_layers * 2 * s.t_kv_heads * 64 * 2 * (L + new_tokens / 2)   # caches hold the key/value heads once (grouped-query attention)
    # four batches in flight: execution contexts of the OCR model (mg_ocr_clone), a host thread + stream each
    import threading
    from markushgrapher_amd.inflight import shared_streams
    sts = shared_streams(torch, eng.mem.device, 4)
    ctxs = [(eng, sts[0])] + [(eng.clone(), sts[i]) for i in range(1, 4)]

    solo_ids = eng.generate(ids, pix, new_tokens)[0].cpu().numpy()
    last = {}

    def work(c, st, reps):
        with torch.cuda.device(st.device), torch.cuda.stream(st):
            for _ in range(reps):
                o = c.generate(ids, pix, new_tokens)[0]
            st.synchronize()
            last[id(c)] = o.cpu().numpy()

    def run_all(reps):
        torch.cuda.synchronize(); t0 = time.time()
        th = [threading.Thread(target=work, args=(c, st, reps)) for c, st in ctxs]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        return time.time() - t0
    run_all(1)
    t4 = run_all(2)
    same4 = all(np.array_equal(v, solo_ids) for v in last.values()) and len(last) == 4
    for c, _ in ctxs[1:]:
        c.close()
    return {"pages_per_s": round(B / tn, 2), "ms_per_batch": round(tn * 1e3, 1), "new_tokens": new_tokens, "batch": B, "prompt_len": L,
            "pages_per_s_4_in_flight": round(8 * B / t4, 2), "ids_4_in_flight_equal_one_context": bool(same4),
            "vision_plus_prefill_ms": round(t1 * 1e3, 2), "decode_step_ms": round(step_ms, 4),
            "dec_hbm_frac": round((wbytes + kvbytes) / (step_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
            "config": "ChemicalOCR stage alone: SmolDocling-256M geometry (INFERRED), recipe weights, one 512-px page per sequence, "
                      "greedy, EOS impossible; vision tower / prefill: second form (tiled fp32 residual streams + batched residual epilogue, tile-wise LayerNorm / RMSNorm, bias-free attention on the encoder's second-form kernel); decode step: 4 launches per layer at one row tile (rotary grouped-query attention + cache append, o_proj + norm, gate/up + SwiGLU, [down_proj + norm | next QKV]), replayed as a HIP graph"}


def configs4_run(eng, B, new_tokens, ocr_pages=128, n_scripts=32, ocr_slots=0, main_inflight=1, ocr_inflight=1, main_batch=None, overlap_slab=0):
    """BASELINE configs[4] measured as ONE loop on one GPU (markushgrapher_amd/pipeline.py): 128 IP5-M-shaped pages (1024 px u8 crops,
    resident) -> device LANCZOS -> ChemicalOCR (SmolDocling-256M geometry, 128 pages per call) -> text -> cells -> tokens -> VTL encoder +
    256-token greedy decode (continuous decoder, 32 slots).  No OCR checkpoint / tokenizer model exists offline: the OCR model's lm_head
    is SCRIPTED (ocr_shapes.scripted_state_dict) so that every page emits a real cell string of 10-120 cells
    that really flows through parse_ocr_string, the word-box splitter and the (stock-class, stand-in vocabulary) tokenizer into the VTL
    model; the OCR rows end at their scripted EOS (an OCR call runs as many steps as its longest page needs)."""
    import dataclasses
    import torch
    from markushgrapher_amd import synth
    from markushgrapher_amd.ocr import OcrEngine
    from markushgrapher_amd.ocr_shapes import PRESETS, script_texts, scripted_state_dict, scripted_prompts, synth_cell_text, detokenize
    from markushgrapher_amd.pipeline import Configs4Pipeline
    from markushgrapher_amd.standin import make_udop_tokenizer
    s = PRESETS["smoldocling"]
    n_cells = synth.randint("configs4/cells", n_scripts, 10, 120, synth.BENCH_SEED)
    texts = [synth_cell_text(int(n), synth.BENCH_SEED, f"p{i}") for i, n in enumerate(n_cells)]
    id_to_piece, chains, starts = script_texts(s, texts)
    ocr = OcrEngine(s).load_state_dict(scripted_state_dict(s, chains, starts))
    prompts = np.concatenate([scripted_prompts(s, chains, starts)] * (ocr_pages // n_scripts), axis=0)
    longest = max(len(c) for c in chains)
    pipe = Configs4Pipeline(ocr, eng, make_udop_tokenizer(), lambda row: detokenize(id_to_piece, row, s.eos_token_id, s.pad_token_id), prompts,
                            ocr_max_new_tokens=longest + 8, max_length=new_tokens + 1, min_length=new_tokens + 1, continuous=True, main_batch=main_batch or B,
                            ocr_slots=ocr_slots, main_inflight=main_inflight, ocr_inflight=ocr_inflight, overlap_slab=overlap_slab)
    if main_inflight > 1:
        pipe.continuous = False          # forced-length decode: one mg_generate per `main_batch` pages and context
    pages = torch.from_numpy(synth.synth_pages_u8(32, 1024, synth.BENCH_SEED)).cuda()
    pages = torch.cat([pages] * (ocr_pages // 32), dim=0)

    def clock():
        torch.cuda.synchronize()
        return time.time()
    pipe(pages)
    t0 = clock()
    res = pipe(pages, timer=clock)
    dt = clock() - t0
    pipe.close()
    ok = sum(res.ocr_texts[i] == texts[i % n_scripts] for i in range(ocr_pages))
    eng.set_padding_semantics(False)             # (the pipeline switched the engine to per-image padding semantics)
    L = res.attention_mask.sum(axis=1)
    return {"pages_per_s": round(ocr_pages / dt, 2), "pages": ocr_pages, "ms_total": round(dt * 1e3, 1),
            "ocr_s": round(res.timings["ocr_s"], 3), "host_s": round(res.timings["host_s"], 3),
            "main_s": round(res.timings["main_s"], 3),
            "vtl_pages_per_call": int(main_batch or B),
            "stages": ((f"OVERLAPPED: OCR of the next slab of {overlap_slab} pages on {ocr_inflight} context(s) with their own streams while the VTL stage's {main_inflight} contexts decode the previous slab (ocr_s = busy time of the OCR worker, main_s = the whole loop)" if res.timings.get("overlapped") else f"one after the other; OCR stage on {ocr_inflight} execution context(s), VTL stage on {main_inflight}")
                       + (" (host stage pipelined with it: main_s contains host_s)" if main_inflight > 1 else "")),
            "ocr_form": (f"queue form, {ocr_slots} decode rows, {res.timings.get('ocr_steps')} steps" if ocr_slots else "batch form: every call walks to its longest page"),
            "ocr_steps_longest_page": longest, "ocr_tokens_mean": round(float(np.mean([len(c) for c in chains])), 1),
            "cells_per_page": [int(n_cells.min()), int(n_cells.max())], "vtl_text_tokens_mean": round(float(L.mean()), 1),
            "vtl_text_tokens_max": int(L.max()), "ocr_strings_as_scripted": f"{ok}/{ocr_pages}", "main_new_tokens": new_tokens,
            "config": f"configs[4] as one loop on one GPU: {ocr_pages} pages per OCR call (SmolDocling-256M geometry, scripted lm_head so that real cell "
                      "strings flow), host text stage (stock UdopTokenizer class, stand-in vocabulary), VTL stage = headline model through the "
                      "continuous decoder (32 slots, forced 256 new tokens); preprocessing, both models and the host stage inside the timed region"}


def ocr_cpu_baseline(B=1, new_tokens=256, sample_steps=16):
    """The reference's own CPU path for this stage is stock transformers on the host (chemical_ocr.py:366-392); what travels to the GPU
    box is its restatement oracle/ocr_oracle.py (fp32 torch-CPU, pinned on stock).  One page, SmolDocling-256M geometry: vision tower
    + prefill once, `sample_steps` decode steps timed and extrapolated to `new_tokens`."""
    import dataclasses
    import torch
    from markushgrapher_amd.ocr_shapes import PRESETS, recipe_state_dict, synth_inputs
    from oracle.ocr_oracle import OcrOracle
    prev = torch.get_num_threads()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))      # B = 1 on a 135 M-parameter text model: more threads only add overhead
    s = dataclasses.replace(PRESETS["smoldocling"], eos_token_id=-1)
    orc = OcrOracle(s, recipe_state_dict(s))
    ids, pix = synth_inputs(s, B)
    with torch.no_grad():
        orc.generate(ids, pix, 2)                                   # warm-up
        t0 = time.time(); orc.generate(ids, pix, 1); t1 = time.time() - t0
        t0 = time.time(); orc.generate(ids, pix, 1 + sample_steps); tn = time.time() - t0
    step = max(tn - t1, 1e-6) / sample_steps
    total = t1 + step * (new_tokens - 1)
    used = torch.get_num_threads()
    torch.set_num_threads(prev)
    return {"pages_per_s": round(B / total, 4), "unit": "pages/s", "cores": used, "kind": "port",
            "sample": f"oracle fp32 torch-CPU, B={B}: vision + prefill {t1:.2f}s + {step * 1e3:.0f} ms/step ({sample_steps} timed steps, extrapolated to {new_tokens} new tokens)"}


def pmc_child(args):
    """Child mode (run under rocprofv3 --pmc FETCH_SIZE by the parent): one short pass of the same workload."""
    import torch
    from markushgrapher_amd import synth
    from markushgrapher_amd.engine import Engine
    shape = synth.SHAPES[args.shape]
    eng = Engine(shape, max_decode_len=64)
    eng.load_state_dict(synth.recipe_state_dict(shape, **synth.BENCH_RECIPE))
    inp = synth.synth_batch(shape, args.batch, seed=synth.BENCH_SEED, return_pages=True)
    nb = max(1, args.pmc_batches)                  # batches in the call, as in the parent's timed calls
    inp = {k: np.concatenate([np.asarray(v)] * nb, axis=0) for k, v in inp.items()}
    pix = eng.preprocess(inp["pages_u8"])
    eng.generate(inp["input_ids"], inp["bbox"], inp["attention_mask"], pix, max_length=args.pmc_child + 1,
                 min_length=args.pmc_child + 1)
    torch.cuda.synchronize()


def pmc_traffic(args, nb=1):
    """HBM bytes from the L2's memory-side read counters: FETCH_SIZE per dispatch (KiB; x2 on gfx950 for wide coalesced
    streams, MI355X_MICROARCH.md 'HBM') of the cross-attention kernel and of a whole decode step, from a child run of
    this script under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (counters in their own run).  None if unavailable."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="mg_pmc_", dir="/tmp")
    steps = 8
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        subprocess.run([exe, "--pmc", "FETCH_SIZE", "--kernel-trace", "-d", tmp, "-o", "pmc", "--", sys.executable,
                        os.path.abspath(__file__), "--pmc-child", str(steps), "--shape", args.shape, "--batch", str(args.batch),
                        "--pmc-batches", str(nb)],
                       cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=420, check=False)
        import sqlite3
        db = None
        for dp, _, fs in os.walk(tmp):
            for f in fs:
                if f.endswith(".db"):
                    db = os.path.join(dp, f)
        if db is None:
            return None
        rows = sqlite3.connect(db).execute(
            "select kernel_name, value from counters_collection where counter_name = 'FETCH_SIZE'").fetchall()
        xa = [v for n, v in rows if "attn_step_kernel<1, 8, true" in n or "xattn_stream_kernel" in n]
        xa = [v for v in xa if v > 0.5 * max(xa)] if xa else xa          # (the launches of the call itself)
        dec = [v for n, v in rows if any(k in n for k in ("attn_step_kernel", "gemm_rows", "greedy_select", "embed_norm_rows", "xattn_stream_kernel",
                                                         "xq_expand_kernel", "xctx_contract_kernel"))]
        if not xa or not dec:
            return None
        return {"cross_attention_bytes_per_launch": int(sum(xa) / len(xa) * 1024 * 2),
                "decode_step_bytes": int(sum(dec) / steps * 1024 * 2),
                "source": f"rocprofv3 --pmc FETCH_SIZE --kernel-trace on a child run of bench.py ({steps} decode steps, one call of {nb} batch(es)): "
                          "FETCH_SIZE KiB x 1024 x 2 (gfx950 reports half the bytes of wide coalesced reads; other access "
                          "widths uncalibrated)"}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def launcher_selftest(args, world, rank):
    """MG_BENCH_BACKEND=gloo (tests only, no GPU, NOT a measurement): the rank set-up of the multi-GPU run - environment from
    torch.distributed.run, process group, barrier, max-over-ranks reduction, the id exchange's all-gather on a stub payload - and one JSON
    line from rank 0 marked as a self-test."""
    import torch
    import torch.distributed as dist
    from markushgrapher_amd.dist import IdExchange
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = args.batch
    ex = IdExchange(B, torch.device("cpu"), pad_token_id=0)
    ids = torch.full((B, 8), rank + 1, dtype=torch.int64)
    dist.barrier()
    t0 = time.time()
    all_ids, all_len = ex.wait(ex.post(ids))
    dist.barrier()
    t = torch.tensor([time.time() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = bool(all((all_ids[r * B:(r + 1) * B, :8] == r + 1).all() for r in range(world)))
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "backend": "gloo", "exchange_ok": ok, "rows_gathered": int(all_ids.shape[0]),
                          "max_over_ranks_s": round(float(t.item()), 4)}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--repeats", type=int, default=3,
                    help="the timed region of exactly --steps steps is run this many times, each bracketed by barrier + synchronize; "
                         "`value` / `ms_per_step` are the MEDIAN region (value_min / value_max / ms_per_step_all beside them)")
    ap.add_argument("--shape", default="large")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--new-tokens", type=int, default=256)
    ap.add_argument("--beams", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-runs", action="store_true", help="skip the EOS-enabled and beam-5 side measurements")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 FETCH_SIZE child pass")
    ap.add_argument("--pmc-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-batches", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--profile-every", type=int, default=16)
    ap.add_argument("--decode-graph", type=int, default=1, help="1: replay the captured decode-step HIP graph; 0: eager launches")
    ap.add_argument("--inflight", type=int, default=4,
                    help="batches in flight per GPU: execution contexts (mg_clone) with a stream and host thread each; 1 = one batch "
                         "after the other, as the reference's loop.  (The id exchange - a few small kernels per batch on the null "
                         "stream / RCCL's stream - is a fifth queue in use for microseconds at a time; measured harmless at one GPU.)")
    ap.add_argument("--batches-per-call", type=int, default=5,
                    help="at most that many batches of `--batch` images (<= 8: 256 rows, the C ABI's limit; 5 is the fastest) in ONE generate call of a context (their rows side "
                         "by side in the decode step: the decoder's weights are read once per step for all of them); the `--steps` batches are "
                         "cut into near-equal calls so that every context is busy to the end; every image's ids are bit-identical to a call "
                         "on its batch alone (tests/test_engine.py::test_rows_do_not_depend_on_the_row_count)")
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)
    # `python bench.py --gpus N` started as ONE process (no torch.distributed.run around it): start the N ranks here - the same
    # command under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on the loopback address - and pass its output
    # and exit code through.  Under torch.distributed.run (WORLD_SIZE set) this is skipped: the process is a rank.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        return sys.exit(subprocess.call(cmd, env=env))

    import torch
    from markushgrapher_amd import synth
    from markushgrapher_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but torch.distributed.run started {world} processes (WORLD_SIZE={world})")
    backend = os.environ.get("MG_BENCH_BACKEND", "nccl")      # "gloo": launcher / rendezvous test without GPUs (tests/test_dist.py); never a measurement
    if backend == "gloo":
        return launcher_selftest(args, world, rank)
    torch.cuda.set_device(local_rank)
    dist = None
    # MG_BENCH_FORCE_DIST=1: a process group of ONE rank over RCCL, the id exchange issued as a real all-gather - the multi-GPU code path
    # (group set-up, asynchronous collectives beside the contexts in flight, barrier) on a single GPU
    force_dist = os.environ.get("MG_BENCH_FORCE_DIST", "0") in ("1", "2")          # 2: the group only, the exchange stays a copy
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))

    shape = synth.SHAPES[args.shape]
    B, new_tokens = args.batch, args.new_tokens
    max_length = new_tokens + 1
    t0 = time.time()
    sd = synth.recipe_state_dict(shape, **synth.BENCH_RECIPE)
    t_weights = time.time() - t0
    eng = Engine(shape, max_decode_len=512)
    eng.load_state_dict(sd)
    eng.set_decode_graph(args.decode_graph)
    # each rank gets its own shard of the global batch (independent images, no data-path exchange)
    # DIFFERENT images in every batch of the timed region: batch j of this rank is drawn with its own seed (j = 0: the batch the parity
    # fixture tests/golden/g4_bench.npz pins), all batches padded to the pool's longest text (ids 0 / box 0 / mask 0, as the reference's
    # collator pads: ref core/trainers/data_collator.py:55-108), so that a call really holds 160 different images at a common L_max
    n_pool = max(1, min(args.steps, int(os.environ.get("MG_BENCH_DISTINCT", "20"))))
    pool_np = [synth.synth_batch(shape, B, seed=synth.BENCH_SEED + rank + 1000 * j, return_pages=True) for j in range(n_pool)]
    L = max(p_["input_ids"].shape[1] for p_ in pool_np)
    for p_ in pool_np:
        padn = L - p_["input_ids"].shape[1]
        if padn:
            p_["input_ids"] = np.pad(p_["input_ids"], ((0, 0), (0, padn)))
            p_["attention_mask"] = np.pad(p_["attention_mask"], ((0, 0), (0, padn)))
            p_["bbox"] = np.pad(p_["bbox"], ((0, 0), (0, padn), (0, 0)))
    dtypes = {"input_ids": np.int64, "bbox": np.float32, "attention_mask": np.uint8, "pixel_values": np.float32, "pages_u8": np.uint8}
    pool = [{k: eng.mem.asarray(v, dtypes[k]) for k, v in p_.items() if k != "pixel_values"} for p_ in pool_np]
    inp, dev = pool_np[0], pool[0]
    # the exchange: static [B, 512] int32 ids + [B] int32 lengths per rank, posted asynchronously and
    # double-buffered (markushgrapher_amd/dist.py), so the gather of batch i overlaps the encoder of batch i+1
    # (at world == 1 the exchange degenerates to its single-rank copy path, which runs all the same: the packing of the static
    #  [32, 512] int32 block + lengths is part of the step on every node size)
    from markushgrapher_amd.dist import IdExchange
    ex = IdExchange(B, torch.device("cuda", local_rank), pad_token_id=shape.pad_token_id, always_collective=os.environ.get("MG_BENCH_FORCE_DIST") == "1")
    handles = []

    def step(beams=args.beams, max_len=max_length, min_len=max_length):
        # the 1024 px u8 crops are what is resident in HBM: LANCZOS resize to the 512 px model input + normalisation run
        # on the device inside the step (bit-exact with the reference's Pillow preprocessing)
        pix = eng.preprocess(dev["pages_u8"])
        ids, _, _ = eng.generate(dev["input_ids"], dev["bbox"], dev["attention_mask"], pix,
                                 num_beams=beams, max_length=max_len, min_length=min_len)
        if ex is not None:
            handles.append(ex.post(ids))
            if len(handles) > 1:
                ex.wait(handles.pop(0))      # the previous batch's gather has had this batch's whole step to complete
        return ids

    # batches in flight: `--inflight` execution contexts on the same weights (include/mgrapher.h mg_clone), one host thread and stream
    # each.  Every step is still one whole pass preprocess -> encoder -> 256 decode steps over one batch of 32; the decode step of one
    # batch is latency-bound for five of its six launches per layer, the others' launches fill the machine it leaves idle.  Results
    # per batch are identical to a call made alone (tests/test_inflight.py).
    from markushgrapher_amd.inflight import InFlight
    fl = InFlight(eng, max(1, args.inflight))
    # Cross-attention form (mg_set_cross_absorb): the library's default picks it by the call's decode rows (weight-absorbed from 96 rows on).
    # The timed region's calls hold bpc x 32 rows: their form is PINNED on every context, so that the per-batch identity check below (a batch
    # of a packed call against a call on the batch alone) compares like with like; the one-batch and small-queue side runs go back to the default.
    bpc_ = max(1, min(8, args.batches_per_call)) if args.beams == 1 else 1
    pin_absorb = args.beams == 1 and bpc_ * B >= eng.ABSORB_AUTO_ROWS and eng.cross_absorb == "auto"
    if pin_absorb:
        for c_ in fl.contexts:
            c_.set_cross_absorb(True)

    # up to `--batches-per-call` batches ride in one call (rows [0, B) = one batch, [B, 2B) the next ...): the same preprocess ->
    # encoder -> decode steps per batch, the decode step's weight stream shared by the batches of the call.
    bpc = max(1, min(8, args.batches_per_call)) if args.beams == 1 else 1      # (default 5: calls of 6 / 8 batches measured slower, profiles/r04_o_batches_per_call.txt)
    call_cache = {}

    def call_inputs(first, nb):
        # batches first .. first + nb - 1 of the pool (cyclic) side by side; built once per (first, nb), outside the timed region
        key = (first % n_pool, nb)
        if key not in call_cache:
            parts = [pool[(first + i) % n_pool] for i in range(nb)]
            call_cache[key] = parts[0] if nb == 1 else {k: torch.cat([q[k] for q in parts], dim=0) for k in parts[0]}
        return call_cache[key]

    from markushgrapher_amd.inflight import plan_calls as _plan_calls

    def plan_calls(k):
        # k batches over the contexts in near-equal calls of at most `bpc` batches (markushgrapher_amd/inflight.py): every context stays
        # busy to the end (20 batches on 4 contexts: a call of 3 and a call of 2 each)
        return _plan_calls(k, len(fl), bpc)

    calls_on_first = []          # batches per call of the calls that ran on the first context (the one the phase events are read from)

    def job(ctx, nb, first=0):
        src = call_inputs(first, nb)
        pix = ctx.preprocess(src["pages_u8"])
        out, _, _ = ctx.generate(src["input_ids"], src["bbox"], src["attention_mask"], pix, num_beams=args.beams,
                                 max_length=max_length, min_length=max_length)
        if ctx is eng:
            calls_on_first.append(nb)
        return out

    last_call = [None]

    def run_calls(sizes):
        firsts = [int(sum(sizes[:i])) for i in range(len(sizes))]
        for nb, first in zip(sizes, firsts):
            call_inputs(first, nb)
        futs = [fl.submit(job, nb, first) for nb, first in zip(sizes, firsts)]
        out = None
        for f, nb, first in zip(futs, sizes, firsts):
            res = f.result()
            last_call[0] = (res, first)
            for j in range(res.shape[0] // B):          # one exchange per batch, as with one batch per call
                out = res[j * B:(j + 1) * B]
                handles.append(ex.post(out))
                if len(handles) > 1:
                    ex.wait(handles.pop(0))      # the previous batch's gather has had this batch's whole step to complete
        return out

    # warm-up: every call size of the timed plan once on every context (graph capture per shape; len(fl) long jobs submitted together land
    # on len(fl) different contexts), then whole plans until at least `warmup` batches have run
    timed_plan = plan_calls(args.steps)
    torch.cuda.synchronize()
    if args.warmup:
        warmed = 0
        for nb_ in sorted(set(timed_plan)):
            run_calls([nb_] * len(fl))
            warmed += nb_ * len(fl)
        while warmed < args.warmup:
            run_calls(timed_plan)
            warmed += args.steps
    while handles:
        ex.wait(handles.pop(0))
    # live timing of the dominant kernel on the launch stream (HIP events), sampled every N-th decode step; phase events
    L_ = eng.lib
    L_.mg_profile_cross_attention.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L_.mg_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L_.mg_profile_read_overhead.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L_.mg_profile_phases.argtypes = [C.c_void_p, C.c_int]
    L_.mg_profile_phases_read.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    nl = shape.num_decoder_layers

    def profile_on():
        L_.mg_profile_cross_attention(eng.model, args.profile_every, (new_tokens // max(args.profile_every, 1) + 1) * nl)
        L_.mg_profile_phases(eng.model, 1)

    def profile_read():
        n_l, ms, keys, empty_ms = C.c_long(0), C.c_double(0), C.c_double(0), C.c_double(0)
        L_.mg_profile_read(eng.model, C.byref(n_l), C.byref(ms), C.byref(keys))
        L_.mg_profile_read_overhead(eng.model, C.byref(empty_ms))
        n_ph, enc_ms, dec_ms = C.c_long(0), C.c_double(0), C.c_double(0)
        L_.mg_profile_phases_read(eng.model, C.byref(n_ph), C.byref(enc_ms), C.byref(dec_ms))
        L_.mg_profile_cross_attention(eng.model, 0, 0)
        L_.mg_profile_phases(eng.model, 0)
        return n_l, ms, keys, empty_ms, n_ph, enc_ms, dec_ms

    profile_on()                       # on the first context: its launches are bracketed while the other contexts run beside it
    import resource
    import threading
    repeats = max(1, args.repeats)
    dts, cpu_s = [], []
    for _rep in range(repeats):        # every repeat: EXACTLY --steps steps between barrier + synchronize on both sides, max over ranks
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.time()
        del calls_on_first[:]
        ids = run_calls(timed_plan)
        nb_first = float(np.mean(calls_on_first)) if calls_on_first else float(np.mean(timed_plan))
        while handles:
            all_ids, all_len = ex.wait(handles.pop(0))      # the last batch's exchange completes inside the timed region
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.time() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        dts.append(dt)
        cpu_s.append((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime))
    dt = float(np.median(dts))         # `value` = the median region
    host_threads = threading.active_count()
    try:
        with open("/proc/self/status") as f_:
            host_threads_os = int([ln.split()[1] for ln in f_ if ln.startswith("Threads:")][0])
    except Exception:
        host_threads_os = None
    n_l, ms, keys, empty_ms, n_ph, enc_ms, dec_ms = profile_read()
    # the same step with ONE batch in flight (the reference's loop shape; rounds 1-2 measured this): untimed for `value`, it gives
    # the kernel's and the phases' uncontended figures beside the in-flight ones
    solo = None
    solo_call = None
    ids_call, first_call = last_call[0]
    ids_equal_solo = None
    if (len(fl) > 1 or bpc > 1) and rank == 0:
        eng.set_shared_gpu(False)        # the context runs alone from here on (InFlight had set it: mg_set_shared_gpu)
        # every batch of the last timed call against ONE call on that batch alone (one context, one batch per call, same padded length)
        eq = []
        for j in range(ids_call.shape[0] // B):
            one = job(eng, 1, first_call + j)
            eq.append(bool(torch.equal(ids_call[j * B:(j + 1) * B], one)))
        ids_equal_solo = bool(all(eq))
        if pin_absorb:
            eng.set_cross_absorb("auto")         # one batch of 32 rows per call: the library's default form for that call size (K / V streams)
        step()
        profile_on()
        torch.cuda.synchronize(); ts = time.time()
        for _ in range(SOLO_STEPS):
            step()
        while handles:
            ex.wait(handles.pop(0))
        torch.cuda.synchronize(); ts = time.time() - ts
        solo = (ts,) + profile_read()
        if pin_absorb:
            eng.set_cross_absorb(True)
        # ... and ONE call of the timed region's largest call shape alone on the first context: the dominant launch and the phases of
        # that shape without other contexts' kernels beside them
        nb_main = max(timed_plan) if timed_plan else 1
        if nb_main > 1:
            def call_alone():
                res = job(eng, nb_main)
                for j in range(res.shape[0] // B):
                    handles.append(ex.post(res[j * B:(j + 1) * B]))
                    if len(handles) > 1:
                        ex.wait(handles.pop(0))
            with torch.cuda.stream(fl.streams[0]):
                call_alone()
                profile_on()
                torch.cuda.synchronize(); tc = time.time()
                for _ in range(2):
                    call_alone()
                while handles:
                    ex.wait(handles.pop(0))
                torch.cuda.synchronize(); tc = time.time() - tc
            solo_call = (tc, nb_main) + profile_read()
        eng.set_shared_gpu(len(fl) > 1)
    assert ids.shape == (B, max_length), ids.shape
    assert ids_equal_solo is not False, "ids of a batch inside a multi-batch call differ from the call on the batch alone"
    # The reference's shipped architecture (config/predict.yaml: architecture_variant me-lf-stack-1): the OCSR vision branch (Swin-B at
    # 384 px + MLP projector, csrc/swin.hip) attached to every context, evaluated inside the timed step from the step's own pixel_values,
    # the decoder cross-attending over [144 e1 tokens | VTL states].  Same plan, same inputs, same clock as the headline region.
    e1_run = None
    if rank == 0 and world == 1 and args.beams == 1 and args.shape == "large" and not args.no_extra_runs and os.environ.get("MG_BENCH_E1", "1") != "0":
        from markushgrapher_amd.e1 import E1Engine
        from markushgrapher_amd.e1_shapes import PRESETS as E1_PRESETS, recipe_state_dict as e1_recipe
        s1 = E1_PRESETS["swin_b_384"]
        e1e = E1Engine(s1).load_state_dict(e1_recipe(s1))
        ctx_all = [eng] + [c for c in fl.contexts if c is not eng]
        for c in ctx_all:
            c.attach_e1(e1e)
        for nb_ in sorted(set(timed_plan)):
            run_calls([nb_] * len(fl))
        while handles:
            ex.wait(handles.pop(0))
        L_.mg_profile_phases(eng.model, 1)
        torch.cuda.synchronize(); t1 = time.time()
        run_calls(timed_plan)
        while handles:
            ex.wait(handles.pop(0))
        torch.cuda.synchronize(); t1 = time.time() - t1
        n_ph1, enc_ms1, dec_ms1 = C.c_long(0), C.c_double(0), C.c_double(0)
        L_.mg_profile_phases_read(eng.model, C.byref(n_ph1), C.byref(enc_ms1), C.byref(dec_ms1))
        L_.mg_profile_phases(eng.model, 0)
        # the branch alone: one call shape of the timed plan (nb batches of 32) and one batch, on the first context's stream
        nb_main = max(timed_plan) if timed_plan else 1
        pix_nb = eng.preprocess(call_inputs(0, nb_main)["pages_u8"])
        alone = {}
        for nb_, px in ((nb_main, pix_nb), (1, pix_nb[:B])):
            e1e.encode(px)
            torch.cuda.synchronize(); ta = time.time()
            for _ in range(3):
                e1e.encode(px)
            torch.cuda.synchronize()
            alone[nb_] = (time.time() - ta) / 3
        from tools.e1_bench import flops_per_image as e1_flops
        fpi = e1_flops(s1)
        e1_run = {"images_per_s": round(B * args.steps / t1, 2), "ms_per_step": round(t1 / args.steps * 1e3, 2), "steps": args.steps,
                  "e1_tokens_per_image": e1e.out_tokens, "e1_gflop_per_image": round(fpi / 1e9, 1),
                  "e1_alone_ms_per_32_images": round(alone[1] * 1e3, 2), "e1_alone_ms_per_call": round(alone[nb_main] * 1e3, 2), "e1_call_images": nb_main * B,
                  "e1_alone_mfma_frac": round(fpi * nb_main * B / alone[nb_main] / (MFMA_PEAK_TFLOPS * 1e12), 4),
                  "encoder_phase_ms_per_call": round(enc_ms1.value / max(n_ph1.value, 1), 2), "decode_step_ms": round(dec_ms1.value / max(n_ph1.value, 1) / new_tokens, 4),
                  "config": "the headline plan with the OCSR vision branch attached (mg_attach_e1: MolScribe Swin-B geometry 384 px, 86.9 M parameters, recipe weights, "
                            "2-layer GELU projector INFERRED): per batch + bilinear 512 -> 384 resize + Swin-B + projector inside the step, + 144 keys per image in the "
                            "cross-K/V projections and in every decode step's cross-attention stream; the reference's architecture_variant me-lf-stack-1"}
        for c in ctx_all:
            c.attach_e1(None)
        e1e.close()
        del pix_nb
        torch.cuda.empty_cache()

    if rank == 0:
        H, d, dff, V = shape.num_heads, shape.d_model, shape.d_ff, shape.vocab_size
        n_enc, n_dec, P = shape.num_layers, shape.num_decoder_layers, shape.num_patches
        # attended encoder positions per image (what the path computes on; padding excluded from the algorithmic work)
        xl = []
        for q in pool:          # (work per batch below = the mean over the pool's batches: xlen holds 32 x n_pool images, sums are divided by n_pool)
            _, msk = eng.encode(q["input_ids"], q["bbox"], q["attention_mask"], eng.preprocess(q["pages_u8"]))
            xl.append(msk.sum(dim=1).cpu().numpy().astype(np.float64))
        xlen = np.concatenate(xl)
        absorbed = args.beams == 1 and (eng.cross_absorb is True or (eng.cross_absorb == "auto" and int(round(nb_first)) * B >= eng.ABSORB_AUTO_ROWS))
        absorbed_main = absorbed
        traffic = None if (args.no_pmc or world > 1 or args.beams != 1) else pmc_traffic(args, int(round(nb_first)))
        def make_roof(n_l, ms, keys, empty_ms, with_traffic):
            if n_l.value <= 0:
                return None
            survey_bytes = keys.value / n_l.value * H * 64 * 2 * 2         # K and V rows of 64 bf16, all heads
            # weight-absorbed form (mg_set_cross_absorb, the default for greedy calls): the launch streams the attended encoder states
            # themselves, d bf16 per position - the bytes the kernel MOVES; `frac` is on these
            bytes_per_launch = keys.value / n_l.value * d * 2 if absorbed else survey_bytes
            raw_s = ms.value / n_l.value * 1e-3            # e0 -> e1 around the launch
            empty_s = empty_ms.value / n_l.value * 1e-3    # e1 -> e2 with nothing in between: cost of the bracket itself
            # The bracket over-reads the kernel by the dispatch latency behind the first record (rocprofv3 kernel trace of
            # the same command: profiles/); the empty bracket over-corrects, so the conservative raw bracket is `achieved`.
            ach = bytes_per_launch / raw_s / 1e9
            r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
            if with_traffic:
                r.update({"traffic": traffic["cross_attention_bytes_per_launch"] if traffic else None, "traffic_unit": "bytes/launch",
                          "traffic_source": traffic["source"] if traffic else None,
                          "kernel": ("xattn_stream_kernel<16> (decoder cross-attention, weight-absorbed: one stream of the encoder states per layer, "
                                     "scores and context on MFMA)") if absorbed else
                                    "attn_step_kernel<1, 8, true> (decoder cross-attention, single query per image/head)"})
            r.update({"bytes_per_launch": int(bytes_per_launch), "survey_bytes_per_launch": int(survey_bytes),
                      "bytes_note": ("bytes_per_launch = bytes the launch moves (sum of attended positions x 2 d_model: the weight-absorbed form reads the encoder "
                                     "states once per layer); survey_bytes_per_launch = SURVEY.md 8d's K + V formula (4 H d_kv per position), which the form no longer moves; "
                                     "achieved / frac are on bytes_per_launch") if absorbed else "K and V rows of every attended position (SURVEY.md 8d)", "avg_launch_us": round(raw_s * 1e6, 2),
                      "empty_bracket_us": round(empty_s * 1e6, 2), "launches_timed": int(n_l.value)})
            return r

        # F_enc = N_enc*S*(8d^2 + 4*d*dff + 4*S*d) + 2*P*768*d;  F_xkv = N_dec*S_x*4d^2  (per image, S = attended positions)
        f_enc = float(np.sum(n_enc * xlen * (8 * d * d + 4 * d * dff + 4 * xlen * d) + 2 * P * (shape.num_channels * shape.patch_size ** 2) * d)) / n_pool
        # cross-K/V projections: executed only in the K / V form (the absorbed form needs none: the layers read the states themselves)
        f_xkv = 0.0 if absorbed else float(np.sum(n_dec * xlen * 4 * d * d)) / n_pool
        tbar = (new_tokens - 1) / 2.0
        # Bytes_step = 2*(N_dec*16d^2 + d*V) + sum_b 2*N_dec*2*d*(S_x + t);  F_step = N_dec*(12d^2 + 4*d*dff) + N_dec*4*d*(S_x+t) + 2*d*V
        # (absorbed form: one stream of d bf16 per attended position and layer instead of K and V)
        bytes_step = 2.0 * (n_dec * 16 * d * d + d * V) + float(np.sum(n_dec * 2 * d * ((1 if absorbed else 2) * xlen + 2 * tbar))) / n_pool
        f_step = float(np.sum(n_dec * (12 * d * d + 4 * d * dff) + n_dec * 4 * d * (xlen + tbar) + 2 * d * V)) / n_pool

        # a call that holds nb batches streams the weights once per step and every batch's K/V
        bytes_w = 2.0 * (n_dec * 16 * d * d + d * V)

        def bytes_step_call(nb):
            return bytes_w + nb * (bytes_step - bytes_w)

        def make_phases(n_ph, enc_ms, dec_ms, nb=1):
            if n_ph.value <= 0 or args.beams != 1:
                return None
            t_enc = enc_ms.value / n_ph.value * 1e-3
            t_step = dec_ms.value / n_ph.value * 1e-3 / new_tokens
            return {"encoder_ms": round(t_enc * 1e3, 2), "decode_step_ms": round(t_step * 1e3, 4), "batches_per_call": nb,
                    "enc_flops": nb * (f_enc + f_xkv), "enc_mfma_frac": round(nb * (f_enc + f_xkv) / t_enc / (MFMA_PEAK_TFLOPS * 1e12), 4),
                    "dec_bytes_step_algorithmic": int(bytes_step_call(nb)),
                    "dec_hbm_frac": round(bytes_step_call(nb) / t_step / (HBM_PEAK_GBS * 1e9), 4),
                    "dec_mfma_frac": round(nb * f_step / t_step / (MFMA_PEAK_TFLOPS * 1e12), 5)}

        def whole_job(n_batches, seconds, nb=1):
            # algorithmic bytes of all decode steps and algorithmic flops of everything, over the elapsed time of the region
            return {"hbm_frac_decode_bytes": round(n_batches / nb * new_tokens * bytes_step_call(nb) / seconds / (HBM_PEAK_GBS * 1e9), 4),
                    "mfma_frac_all_flops": round(n_batches * (f_enc + f_xkv + new_tokens * f_step) / seconds / (MFMA_PEAK_TFLOPS * 1e12), 4)}

        roof = make_roof(n_l, ms, keys, empty_ms, True)
        if roof is not None:
            roof["timing"] = ("HIP events on the launch stream around the device-counter form of the launch the decode graph "
                              "replays: (record, launch, record, record); avg_launch_us = first bracket, uncorrected; "
                              "empty_bracket_us = second bracket (nothing in between); taken on the first execution context during "
                              "the timed region, i.e. with the other %d contexts' kernels sharing the GPU; a launch covers the %d batch(es) "
                              "of its call" % (len(fl) - 1, int(round(nb_first))))
            roof["batches_in_flight"] = int(sum(timed_plan[:len(fl)]))
        phases = make_phases(n_ph, enc_ms, dec_ms, nb_first)
        if roof is not None and phases is not None:
            # what the memory system delivers while all contexts decode: every context moves its call's step bytes per (contended) step time
            agg = len(fl) * bytes_step_call(nb_first) / (phases["decode_step_ms"] * 1e-3) / 1e9
            roof["all_contexts_decode_GBps"] = round(agg, 1)
            roof["all_contexts_decode_frac"] = round(agg / HBM_PEAK_GBS, 4)
            roof["note"] = ("`achieved` / `frac` are ONE context's dominant launch timed while %d other contexts' kernels share the GPU (and, with mg_set_shared_gpu, "
                            "with one of its workgroups resident per CU): a quarter of the machine's attention, not the kernel's quality - that is "
                            "one_call_alone.roofline (the same launch shape alone).  all_contexts_decode_*: algorithmic decode bytes of all contexts' steps over "
                            "the contended step time = the HBM rate the decode phases sustain together" % (len(fl) - 1))
        if phases is not None:
            phases["dec_bytes_step_fetched"] = traffic["decode_step_bytes"] if traffic else None
            phases["batches_in_flight"] = int(sum(timed_plan[:len(fl)]))
            phases["whole_job"] = whole_job(args.steps, dt, args.steps / max(len(timed_plan), 1))
            phases["note"] = ("phase times: HIP events in mg_generate on the first execution context [preprocess excluded | encoder + "
                              "cross-K/V | decode loop] = one batch's latency while %d batches share the GPU (per-context fractions "
                              "are of the whole GPU's peak; a call holds `batches_per_call` batches: its encoder flops are theirs together, "
                              "its step's algorithmic bytes = the weights once + every batch's K/V); whole_job = algorithmic decode bytes / all algorithmic flops of the K "
                              "batches over the timed region; formulas SURVEY.md §8d with S = attended positions per image (mean "
                              "%.0f), t = mean decode position; the decode step is HBM-bound (dec_mfma_frac is reported because "
                              "north_star asks for it); fetched bytes: child run of 8 steps (t < 8), they contain the product "
                              "weights of the pair projections (+1.37x on weights, DESIGN.md) and not the cross-K/V projection "
                              "weights the formula's 16d^2 counts" % (len(fl), float(xlen.mean())))
        single = None
        if solo is not None:
            ts = solo[0]
            if pin_absorb:
                absorbed = False          # the one-batch leg ran the default form of a 32-row call: K / V streams (accounting below follows)
                f_xkv = float(np.sum(n_dec * xlen * 4 * d * d)) / n_pool
                bytes_step = 2.0 * (n_dec * 16 * d * d + d * V) + float(np.sum(n_dec * 2 * d * (2 * xlen + 2 * tbar))) / n_pool
            single = {"images_per_s": round(B * SOLO_STEPS / ts, 2), "ms_per_batch": round(ts / SOLO_STEPS * 1e3, 2), "steps": SOLO_STEPS,
                      "roofline": make_roof(*solo[1:5], False), "phases": make_phases(*solo[5:8]),
                      "whole_job": whole_job(SOLO_STEPS, ts),
                      "cross_attention_form": "weight-absorbed" if absorbed else "K / V streams (the library's default below 96 decode rows per call)",
                      "note": "the same step with one batch in flight (one context, one stream): the loop shape of the reference and of "
                              "rounds 1-2; kernel and phase figures without other batches' kernels beside them"}
            if pin_absorb:
                absorbed = absorbed_main
                f_xkv = 0.0 if absorbed else f_xkv
                bytes_step = 2.0 * (n_dec * 16 * d * d + d * V) + float(np.sum(n_dec * 2 * d * ((1 if absorbed else 2) * xlen + 2 * tbar))) / n_pool
        call_alone_rep = None
        if solo_call is not None:
            tc, nbm = solo_call[0], solo_call[1]
            call_alone_rep = {"batches_per_call": nbm, "images_per_s": round(B * nbm * 2 / tc, 2), "ms_per_call": round(tc / 2 * 1e3, 2), "calls": 2,
                              "roofline": make_roof(*solo_call[2:6], False), "phases": make_phases(*solo_call[6:9], nbm),
                              "note": "one call of the timed region's largest shape alone on one context (no other contexts' kernels beside it): the "
                                      "dominant launch streams the cross-attention K/V of all its batches in one grid; agrees with the rocprofv3 kernel "
                                      "trace of `bench.py --inflight 1` (profiles/)"}
        extra = None
        if pin_absorb:               # the side runs below (EOS queues, beam search, configs[4]) take the library's default form for their call sizes
            for c_ in fl.contexts:
                c_.set_cross_absorb("auto")
        if not args.no_extra_runs and world == 1 and args.beams == 1:
            extra = {}
            # beam-5, BASELINE configs[2]: 160 live sequences, 128 new tokens
            step(beams=5, max_len=129, min_len=129)
            torch.cuda.synchronize(); tb = time.time()
            step(beams=5, max_len=129, min_len=129)
            torch.cuda.synchronize(); tb = time.time() - tb
            extra["beam5"] = {"images_per_s": round(B / tb, 2), "ms_per_batch": round(tb * 1e3, 1), "new_tokens": 128,
                              "config": "configs[2]: batch 32, num_beams 5 (160 live rows), EOS suppressed; one batch in flight"}
            if len(fl) > 1:
                def job_beam(ctx):
                    out, _, _ = ctx.generate(dev["input_ids"], dev["bbox"], dev["attention_mask"], ctx.preprocess(dev["pages_u8"]),
                                             num_beams=5, max_length=129, min_length=129)
                    return out
                for f in [fl.submit(job_beam) for _ in range(len(fl))]:
                    f.result()
                nb5 = 2 * len(fl)
                torch.cuda.synchronize(); tb = time.time()
                for f in [fl.submit(job_beam) for _ in range(nb5)]:
                    f.result()
                torch.cuda.synchronize(); tb = time.time() - tb
                extra["beam5_in_flight"] = {"images_per_s": round(B * nb5 / tb, 2), "ms_per_batch": round(tb / nb5 * 1e3, 1), "new_tokens": 128,
                                            "batches": nb5, "batches_in_flight": len(fl),
                                            "config": "configs[2] with %d batches in flight (execution contexts as in the headline run)" % len(fl)}
            # EOS enabled (max_length 512): random-init weights never emit EOS on their own, so the EOS row of the tied embedding
            # is scaled up the ladder until at least three quarters of the rows end by themselves; rows then end at different
            # steps, finished rows emit pad and the batch stops when all have ended or at max_length (gen:2927-2937 bookkeeping)
            emb = sd["shared.weight"].copy()
            chosen = None
            for scale in EOS_ROW_SCALES:
                emb[shape.eos_token_id] = synth.round_bf16(sd["shared.weight"][shape.eos_token_id] * np.float32(scale))
                eng.load_state_dict({"shared.weight": emb})
                ie = step(max_len=512, min_len=0).cpu().numpy()
                ended = (ie == shape.eos_token_id).any(axis=1)
                chosen = scale
                if ended.mean() >= 0.75:           # three quarters of the rows end on their own (the rest run to max_length)
                    break
            torch.cuda.synchronize(); te = time.time()
            ids_e = step(max_len=512, min_len=0)
            torch.cuda.synchronize(); te = time.time() - te
            ie = ids_e.cpu().numpy()
            lens = np.array([int(np.argmax(r == shape.eos_token_id)) if (r == shape.eos_token_id).any() else ie.shape[1] - 1 for r in ie])
            extra["eos_enabled"] = {"images_per_s": round(B / te, 2), "ms_per_batch": round(te * 1e3, 1), "max_length": 512,
                                    "decode_steps_run": int(ie.shape[1] - 1), "mean_row_length": round(float(lens.mean()), 1),
                                    "min_row_length": int(lens.min()), "max_row_length": int(lens.max()), "eos_row_scale": chosen,
                                    "config": "greedy, EOS enabled (generate(max_length=512) as the reference calls it), same inputs; EOS "
                                              "embedding row scaled so rows end at different steps; one mg_generate call per 32 images: the batch "
                                              "walks to its longest row"}
            # the same workload through the continuous decoder (mg_generate_stream): a queue of 16 batches' worth of images (the same 32
            # pages repeated - a row's cost depends on its length only), 32 decode slots; a row that ends hands its slot to the next
            # image, the encoder of the next 32 runs ahead on its own stream.  ids per image identical to the batch call (checked here).
            QB = 16
            qd = {k: torch.cat([dev[k]] * QB, dim=0) for k in ("input_ids", "bbox", "attention_mask")}

            def stream_eos():
                pix = torch.cat([eng.preprocess(dev["pages_u8"]) for _ in range(QB)], dim=0)
                return eng.generate_stream(qd["input_ids"], qd["bbox"], qd["attention_mask"], pix, max_length=512, min_length=0,
                                           chunk=B, slots=B, pool_chunks=3)
            stream_eos()
            torch.cuda.synchronize(); ts = time.time()
            ids_s, len_s, steps_s = stream_eos()
            torch.cuda.synchronize(); ts = time.time() - ts
            ids_s, len_s = ids_s.cpu().numpy(), len_s.cpu().numpy()
            same = all(np.array_equal(ids_s[n, :len_s[n]], ie[n % B, :len_s[n]]) for n in range(QB * B))
            extra["eos_enabled_continuous"] = {
                "images_per_s": round(QB * B / ts, 2), "queue_images": QB * B, "slots": B, "decode_steps_run": int(steps_s),
                "mean_row_length": round(float(len_s.mean() - 1), 1), "speedup_vs_batch_calls": round(QB * B / ts / (B / te), 2),
                "ids_equal_batch_calls": bool(same),
                "config": "mg_generate_stream: continuous decoding of a queue of images on 32 slots (finished rows free their slot; encoder + "
                          "cross-K/V of the next 32 images on a second stream), device preprocessing of all pages inside the timed region"}
            if len(fl) > 1:
                # the queue split over the execution contexts, each its own continuous decoder (32 slots; encoder on the context's own
                # stream - the contexts overlap one another, a run-ahead stream per context would only take hardware queues)
                for c in fl.contexts:
                    c.set_stream_encoder(0)
                QF = 2 * QB                                   # a longer queue: each context works through 8 batches' worth (fewer tail steps)
                qf = {k: torch.cat([dev[k]] * QF, dim=0) for k in ("input_ids", "bbox", "attention_mask")}
                per = QF * B // len(fl)

                def job_stream(ctx, i):
                    sl = slice(i * per, (i + 1) * per)
                    pix = torch.cat([ctx.preprocess(dev["pages_u8"]) for _ in range(per // B)], dim=0)
                    o, l, st = ctx.generate_stream(qf["input_ids"][sl], qf["bbox"][sl], qf["attention_mask"][sl], pix, max_length=512,
                                                   min_length=0, chunk=B, slots=2 * B, pool_chunks=4)
                    return o.cpu().numpy(), l.cpu().numpy(), st
                fl.map(job_stream, range(len(fl)))
                torch.cuda.synchronize(); tq = time.time()
                res_q = fl.map(job_stream, range(len(fl)))
                torch.cuda.synchronize(); tq = time.time() - tq
                ids_q = np.concatenate([r[0] for r in res_q]); len_q = np.concatenate([r[1] for r in res_q])
                same_q = all(np.array_equal(ids_q[n, :len_q[n]], ie[n % B, :len_q[n]]) for n in range(QF * B))
                extra["eos_enabled_continuous_in_flight"] = {
                    "images_per_s": round(QF * B / tq, 2), "queue_images": QF * B, "contexts": len(fl), "slots_per_context": 2 * B,
                    "decode_steps_run_per_context": [int(r[2]) for r in res_q], "speedup_vs_batch_calls": round(QF * B / tq / (B / te), 2),
                    "ids_equal_batch_calls": bool(same_q),
                    "config": "a queue of 1024 images cut over the execution contexts of the headline run, one continuous decoder each"}
                for c in fl.contexts:
                    c.set_stream_encoder(1)
            if len(fl) > 1:
                # the reference's shipped decode settings (config/predict.yaml: beam_search True -> num_beams 5; generate(max_length=512), EOS
                # live): one mg_generate per 32 images and context - a batch walks until all of its beams are done (no queue form for beams yet)
                def job_beam_eos(ctx):
                    out, _, _ = ctx.generate(dev["input_ids"], dev["bbox"], dev["attention_mask"], ctx.preprocess(dev["pages_u8"]),
                                             num_beams=5, max_length=512, min_length=0)
                    return out.cpu().numpy()
                for f in [fl.submit(job_beam_eos) for _ in range(len(fl))]:
                    f.result()
                nbe = 2 * len(fl)
                torch.cuda.synchronize(); tbe = time.time()
                outs = [f.result() for f in [fl.submit(job_beam_eos) for _ in range(nbe)]]
                torch.cuda.synchronize(); tbe = time.time() - tbe
                extra["beam5_eos_enabled_in_flight"] = {
                    "images_per_s": round(B * nbe / tbe, 2), "ms_per_batch": round(tbe / nbe * 1e3, 1), "batches": nbe, "batches_in_flight": len(fl),
                    "columns_returned": int(outs[0].shape[1]), "all_batches_equal": bool(all(np.array_equal(o, outs[0]) for o in outs)),
                    "config": "num_beams 5, max_length 512, EOS enabled (EOS row scaled as in eos_enabled): the reference's default decode mode"}
                # the same mode through the beam QUEUE (mg_generate_stream_beam): image slots of 5 rows; an image whose search stops is
                # written out and its slot handed to the next image of the queue - no slot walks to the longest image of a batch
                for c in fl.contexts:
                    c.set_stream_encoder(0)
                QBq = 8                                        # a queue of 8 batches' worth of images per context
                SLOTS_B = 32                                   # 32 image slots x 5 beams = 160 rows, the batch form's row count
                qb = {k: torch.cat([dev[k]] * QBq, dim=0) for k in ("input_ids", "bbox", "attention_mask")}

                def job_beam_queue(ctx, i):
                    pix = torch.cat([ctx.preprocess(dev["pages_u8"]) for _ in range(QBq)], dim=0)
                    o, l, sc, st = ctx.generate_stream_beam(qb["input_ids"], qb["bbox"], qb["attention_mask"], pix, num_beams=5, max_length=512,
                                                            min_length=0, chunk=B, slots=SLOTS_B, pool_chunks=3)
                    return o.cpu().numpy(), l.cpu().numpy(), st
                fl.map(job_beam_queue, range(1))               # (one context: the queue of 256 images alone on the GPU)
                torch.cuda.synchronize(); t1q = time.time()
                (o1, l1, st1), = fl.map(job_beam_queue, range(1))
                torch.cuda.synchronize(); t1q = time.time() - t1q
                ref_b = outs[0]
                same_b = all(np.array_equal(o1[n, :min(int(l1[n]), ref_b.shape[1])], ref_b[n % B, :min(int(l1[n]), ref_b.shape[1])]) for n in range(QBq * B))
                fl.map(job_beam_queue, range(len(fl)))
                torch.cuda.synchronize(); tbq = time.time()
                res_bq = fl.map(job_beam_queue, range(len(fl)))
                torch.cuda.synchronize(); tbq = time.time() - tbq
                extra["beam5_eos_enabled_queue"] = {
                    "images_per_s_one_context": round(QBq * B / t1q, 2), "images_per_s": round(len(fl) * QBq * B / tbq, 2), "contexts": len(fl),
                    "queue_images_per_context": QBq * B, "image_slots": SLOTS_B, "decode_steps_run": int(st1),
                    "decode_steps_run_per_context": [int(r[2]) for r in res_bq], "mean_hypothesis_length": round(float(l1.mean()), 1),
                    "speedup_vs_batch_calls": round(len(fl) * QBq * B / tbq / (B * nbe / tbe), 2), "hypotheses_equal_batch_calls": bool(same_b),
                    "config": "mg_generate_stream_beam: num_beams 5, max_length 512, EOS enabled, 32 image slots of 5 rows working through a "
                              "queue of 256 images per context (finished images free their slot; device preprocessing inside the timed region)"}
                for c in fl.contexts:
                    c.set_stream_encoder(1)
            eng.load_state_dict({"shared.weight": sd["shared.weight"]})
            # the contexts of the runs above keep workspaces sized by their largest call (64 rows x 512 positions: 15 GB each, the queues'
            # K/V pools beside them); the stages below bring their own contexts
            for c in [eng] + list(fl.contexts):
                c.release_workspaces()
            torch.cuda.empty_cache()
            extra["ocr_stage"] = ocr_stage_run()
            if not args.no_cpu_baseline:
                extra["ocr_stage"]["cpu_baseline"] = ocr_cpu_baseline()
            extra["configs4_end_to_end_1gpu_batch_ocr"] = configs4_run(eng, B, new_tokens)
            # the same loop with the OCR stage's queue form and a longer queue (512 pages, 256 OCR decode rows)
            extra["configs4_end_to_end_1gpu_one_context"] = configs4_run(eng, B, new_tokens, ocr_pages=512, ocr_slots=256)
            # the same 512 pages with several batches in flight inside each stage: the OCR stage's pages over 4 execution contexts of the
            # OCR model (128 decode rows each), the VTL stage's batches of 32 over 4 contexts, host stage pipelined with the VTL stage
            # (the VTL calls take two batches of 32 pages each, as the headline run's calls do)
            extra["configs4_end_to_end_1gpu"] = configs4_run(eng, B, new_tokens, ocr_pages=512, ocr_slots=128, main_inflight=4, ocr_inflight=4,
                                                             main_batch=2 * B)
        out = {
            "metric": METRIC, "value": round(world * B * args.steps / dt, 3), "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            # `value` / `ms_per_step` = the median of `repeats` timed regions of exactly `steps` steps each (ms_per_step x steps x repeats of wall clock)
            "repeats": repeats, "value_min": round(world * B * args.steps / max(dts), 3), "value_max": round(world * B * args.steps / min(dts), 3),
            "ms_per_step_all": [round(x / args.steps * 1e3, 2) for x in dts],
            # host side of one rank over a timed region: user + system CPU seconds of this process (all its threads) and its thread count;
            # host_cpu_s / (ms_per_step x steps) = cores one rank keeps busy (what 8 ranks need of the node's cores)
            "host_cpu_s": round(float(np.median(cpu_s)), 3), "host_cpu_cores_busy": round(float(np.median(cpu_s)) / dt, 2),
            "host_threads": host_threads_os, "host_threads_python": host_threads,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[1]: batch 32/GPU synthetic 1024x1024 u8 crops -> device LANCZOS 512px model input, greedy "
                                   f"decode, {new_tokens} forced new tokens (EOS suppressed), UDOP-large-shaped "
                                   "MarkushGrapher-2 VTL encoder + CXSMILES decoder, recipe weights = tests/golden/g4_bench.npz",
                       "ocsr_branch": "not attached: `value` is the VTL-only model (BASELINE.json north_star's path); the reference's shipped "
                                      "me-lf-stack-1 architecture also runs the OCSR vision branch e1 - same plan with it attached: images_per_s_with_e1_branch",
                       "shape": args.shape, "batch_per_gpu": B, "text_len_padded": int(L), "new_tokens": new_tokens,
                       "num_beams": args.beams, "decode_graph": args.decode_graph, "batches_in_flight": int(sum(timed_plan[:len(fl)])),
                       "contexts": len(fl), "batches_per_call_max": bpc, "batches_per_call": timed_plan,
                       "ids_equal_one_batch_calls": ids_equal_solo, "distinct_batches": n_pool,
                       "cross_attention_form": ("weight-absorbed (mg_set_cross_absorb pinned on the contexts: calls of %d decode rows)" % (bpc * B)) if absorbed_main
                                               else "K / V streams",
                       "inputs": f"{n_pool} different batches of 32 images per rank (seed + 1000 j; j = 0 is the batch of tests/golden/g4_bench.npz), padded to the "
                                 f"pool's longest text ({int(L)} tokens): the calls of the timed region hold different images in every row",
                       "single_rank_rccl_group": bool(force_dist),
                       "in_flight": "execution contexts on one set of weights (mg_clone), a stream + host thread + workspace each; a call "
                                    "of a context takes `batches_per_call` batches of 32 (rows side by side: one pass over the decoder's "
                                    "weights per step for all of them); every step is one whole batch start to end; ids identical to "
                                    "one-batch-at-a-time calls (checked in this run: ids_equal_one_batch_calls); warm-up = `warmup` "
                                    "calls per context",
                       "parallelism": f"dp{world} (independent image shards; one RCCL all-gather of [32,512] int32 ids + lengths per batch)"},
            # the three readings of "bs = 32 per GPU" side by side at the top level: `value` = 32-image batches packed into calls of
            # `rows_per_decode_step` rows on `contexts` contexts; one batch per call on one context; the dominant kernel alone
            "images_per_s_one_batch_in_flight": single["images_per_s"] if single else None,
            "rows_per_decode_step": int(B * max(timed_plan)) if timed_plan else B,
            "kernel_alone_frac": (call_alone_rep or {}).get("roofline", {}).get("frac") if call_alone_rep and call_alone_rep.get("roofline") else
                                 ((single or {}).get("roofline") or {}).get("frac"),
            "images_per_s_with_e1_branch": e1_run["images_per_s"] if e1_run else None,
            # the decode mode the reference ships (config/predict.yaml:13 beam search, 5 beams; utils_evaluation.py:278-281 max_length 512, EOS live),
            # as a queue of images over the 4 contexts (mg_generate_stream_beam): extra_runs.beam5_eos_enabled_queue
            "images_per_s_reference_default_mode": ((extra or {}).get("beam5_eos_enabled_queue") or {}).get("images_per_s"),
            "roofline": roof, "phases": phases, "one_call_alone": call_alone_rep, "one_batch_in_flight": single, "with_e1_branch": e1_run,
            "extra_runs": extra,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(shape, sd)
        else:
            out["cpu_baseline"] = None
        out["setup_s"] = {"recipe_weights": round(t_weights, 1)}
        try:
            C.CDLL(None).fflush(None)        # RCCL's version banner sits in the C runtime's stdout buffer: out with it BEFORE the result line
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
