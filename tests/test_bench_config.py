"""Parity of the BENCHMARK configuration itself (BASELINE.json configs[1] greedy and configs[2] beam-5): UDOP-large
shape, B = 32 on bench.py's own inputs (synthetic 1024 px pages -> device LANCZOS -> 512 px), recipe weights with the
bench's gain, against tests/golden/g4_bench.npz minted from stock transformers UDOP (tools/make_golden.py g4; the
reference's fork is unavailable: fork-only pieces stay parity-unpinned).  These are the tests that put the
M = 32 production-dimension decode kernels (and the 160-row beam forms) under an oracle check.

Tolerances (measured with tools/g4_probe.py on MI355X - profiles/r02_g4_probe.log - then set with >= 2x margin):
  ENC_*      encoder output rows (unit-RMS after the final norm x gains in [0.75, 1.25]); measured max 0.025, mean 0.0038,
             per-image |sum| within 8e-6
  LOGIT_TOL  pre-argmax logits (max |logit| 1.28 with this recipe): bf16 operands / fp32 accumulation vs the fp32 reference;
             measured max 0.0089 over 32 images x (16 decode steps + 32 teacher-forced positions) x top-8
  ids        bit-exact wherever the reference's top-1/top-2 margin exceeds MARGIN_TOL = 2 x LOGIT_TOL (two logits each within
             LOGIT_TOL cannot swap across a larger gap); measured: argmax equal in 100 % of the cases with margin > 0.02
"""
import numpy as np
import pytest

from markushgrapher_amd import synth
from tests.backends import make_engine
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu

ENC_MAX, ENC_MEAN, ENC_SUM_REL = 0.06, 0.01, 1e-3
LOGIT_TOL = 0.02
MARGIN_TOL = 2 * LOGIT_TOL

_state = {}


def _setup():
    if not _state:
        g = load_golden("g4_bench.npz")
        shape = synth.SHAPES["large"]
        rec = dict(zip(("gain", "embed_gain", "ffn_gain", "xq_gain"), [float(v) for v in g["recipe"]]))
        assert rec == synth.BENCH_RECIPE and int(g["synth_seed"]) == synth.BENCH_SEED      # the fixture IS bench.py's configuration
        sd = synth.recipe_state_dict(shape, **rec)
        eng = make_engine("hip", shape, sd, max_decode_len=64)
        # the benchmark's calls hold 160 decode rows and run the weight-absorbed cross-attention (the default picks the form by the call's
        # rows): the fixture tests pin that form on their 32-row calls, so that the stock-pinned parity covers the kernels the headline runs
        eng.set_cross_absorb(True)
        inp = synth.synth_batch(shape, int(g["batch"]), seed=int(g["synth_seed"]), return_pages=True)
        pix = eng.preprocess(inp["pages_u8"])              # the bench step's own first stage
        _state.update(g=g, shape=shape, eng=eng, args=(inp["input_ids"], inp["bbox"], inp["attention_mask"], pix))
    return _state["g"], _state["shape"], _state["eng"], _state["args"]


def _check_top8(logits, vals, idx, live=None):
    """logits [B, T, V] (ours) vs the reference's top-8 values / indices per position: every one of the 8 values within
    LOGIT_TOL at the reference's index; the argmax equal wherever the reference margin exceeds MARGIN_TOL; our own top-8
    set equals the reference's wherever the 8th/9th boundary is decided by more than the tolerance can move."""
    at = np.take_along_axis(logits, idx, -1)
    err = np.abs(at - vals)
    if live is not None:
        err = err[live]
    assert err.max() < LOGIT_TOL, err.max()
    am = logits.argmax(-1)
    margin = vals[..., 0] - vals[..., 1]
    sel = margin > MARGIN_TOL
    if live is not None:
        sel &= live
    assert sel.sum() > 0.25 * sel.size
    assert np.array_equal(am[sel], idx[..., 0][sel])
    # rank-k index equal wherever both neighbouring gaps exceed the margin tolerance
    ours = np.argsort(-logits, axis=-1, kind="stable")[..., :8]
    gap_lo = np.concatenate([np.full(vals[..., :1].shape, np.inf), vals[..., :-1] - vals[..., 1:]], -1)
    gap_hi = np.concatenate([vals[..., :-1] - vals[..., 1:], np.zeros(vals[..., :1].shape)], -1)
    solid = (gap_lo > MARGIN_TOL) & (gap_hi > MARGIN_TOL)
    if live is not None:
        solid &= live[..., None]
    assert np.array_equal(ours[solid], idx[solid])
    return err


def test_g4_encoder_all_32_images():
    g, shape, eng, args = _setup()
    enc, mask = eng.encode(*args)
    enc, mask = eng.mem.numpy(enc), eng.mem.numpy(mask)
    assert np.array_equal(mask, g["enc_mask"].astype(np.uint8))
    for b in range(enc.shape[0]):
        v = mask[b].astype(bool)
        s_abs = np.abs(enc[b][v]).astype(np.float64).sum()
        assert abs(s_abs - g["enc_abs_sum"][b]) / g["enc_abs_sum"][b] < ENC_SUM_REL, b
        err = np.abs(enc[b][g["enc_rows"][b]] - g["enc_probe"][b])
        assert err.max() < ENC_MAX and err.mean() < ENC_MEAN, (b, err.max(), err.mean())


def test_g4_encoder_is_bitwise_reproducible():
    """Three passes over the same batch must agree bit for bit (fixed-order reductions, no races in the hand-pipelined
    LDS-DMA loops: a missed wait shows up here as differing bits long before it shows up in a tolerance)."""
    g, shape, eng, args = _setup()
    ref = None
    for _ in range(3):
        enc, _ = eng.encode(*args)
        cur = eng.mem.numpy(enc).copy()
        if ref is None:
            ref = cur
        assert np.array_equal(ref, cur)


def test_g4_greedy_free_running_ids_under_margin_rule():
    """generate() exactly as bench.py calls it (B = 32, EOS suppressed), first 16 steps: ids equal the reference's up to the
    first step of each row whose reference margin is below MARGIN_TOL; per-step top-1 logit within LOGIT_TOL while equal."""
    g, shape, eng, args = _setup()
    new = int(g["new_tokens"])
    ids, _, top2 = eng.generate(*args, max_length=new + 1, min_length=new + 1, return_top2=True)
    ids, top2 = eng.mem.numpy(ids), eng.mem.numpy(top2)
    ref, vals = g["greedy_ids"], g["step_top_vals"]
    margin = vals[..., 0] - vals[..., 1]
    compared = 0
    for b in range(ref.shape[0]):
        for t in range(1, new + 1):
            if margin[b, t - 1] < MARGIN_TOL:
                break
            assert ids[b, t] == ref[b, t], (b, t)
            assert abs(top2[t, b, 0] - vals[b, t - 1, 0]) < LOGIT_TOL
            compared += 1
    # What the free-running rule covers, stated: a row stops at its FIRST step whose stock margin is below MARGIN_TOL, and with
    # random-init weights (median margin 0.04) that comes early - 53 of the 32 x 16 = 512 positions (10 %) are compared here, every
    # one of them must match.  The other 90 % are not skipped by the suite: test_g4_decode_path_teacher_forced_top8 below compares
    # all 512 positions (top-8 logits and ranks) with stock's ids forced in, so that one near-tie does not hide the later steps.
    expected = sum(int(np.argmax(np.append(margin[b] < MARGIN_TOL, True))) for b in range(ref.shape[0]))
    assert compared == expected and compared >= 50, (compared, expected)
    # non-degenerate: many different tokens, image-dependent sequences (the reference's ids: 0 repeats of one token)
    assert len({tuple(r) for r in ref.tolist()}) >= 16 and len(set(ref[:, 1:].ravel().tolist())) > 50
    same_rows = int((ids == ref).all(1).sum())
    print(f"greedy, free-running: {same_rows} of 32 rows identical to stock over all 16 steps")
    assert same_rows >= 19, same_rows         # measured 23 (K / V form) and 22 (weight-absorbed form) of 32 rows identical over all 16 steps; the others part at margins < 0.004; threshold = measured - 3


def test_g4_decode_path_teacher_forced_top8():
    """The KV-cached decode path with the reference's own greedy ids forced in (mg_debug_decode_capture): every one of the
    16 steps of all 32 rows is comparable - logits at the reference's top-8 within LOGIT_TOL, ranks under the margin rule.
    This is the check that covers gemm_rows_*<M=32>, attn_step_kernel<1,8,*> with 32 ragged cross lengths, lm_head."""
    g, shape, eng, args = _setup()
    new, B = int(g["new_tokens"]), int(g["batch"])
    cap = eng.debug_decode_capture(new, B, g["greedy_ids"])
    try:
        ids, _, _ = eng.generate(*args, max_length=new + 1, min_length=new + 1)
        logits = eng.mem.numpy(cap).transpose(1, 0, 2).copy()
    finally:
        eng.debug_decode_capture()
    _check_top8(logits, g["step_top_vals"], g["step_top_idx"])
    # and the forced run's own selections agree with the captured logits (EOS masked out)
    lm = logits.copy()
    lm[..., shape.eos_token_id] = -np.inf
    assert np.array_equal(eng.mem.numpy(ids)[:, 1:], lm.argmax(-1))


def test_g4_teacher_forced_forward_top8_t32():
    """forward() surface at T = 32 decoder positions on the 32 bench images (labels with -100 tails on two rows)."""
    from oracle.udop_oracle import Oracle
    g, shape, eng, args = _setup()
    labels = g["labels"]
    dec_ids = Oracle.shift_right(labels, shape.decoder_start_token_id, shape.pad_token_id).numpy()
    dam = (labels != -100).astype(np.uint8)
    logits, _, _ = eng.forward_logits(*args, dec_ids, dam)
    logits = eng.mem.numpy(logits)
    live = labels != -100
    _check_top8(logits, g["tf_top_vals"], g["tf_top_idx"], live)
    # per-token negative log-likelihood of the labels (what the reference's loss averages, stock:1565-1570)
    lse = np.log(np.exp(logits - logits.max(-1, keepdims=True)).sum(-1)) + logits.max(-1)
    nll = lse - np.take_along_axis(logits, np.clip(labels, 0, None)[..., None], -1)[..., 0]
    assert np.abs(nll - g["tf_nll"])[live].max() < 2 * LOGIT_TOL


def test_g4_beam5_subset_and_full_batch():
    """configs[2]: beam-5.  The 4-image subset against stock's ids / sequence scores; then all 32 images (160 live rows: the
    5-row-tile forms of the decode kernels, whose K-split differs from the one-tile forms, so near-tied beams may part) against
    the same reference scores, and deterministic."""
    g, shape, eng, args = _setup()
    new, nb = int(g["new_tokens"]), int(g["beam_rows"])
    sub = tuple(a[:nb] for a in args)
    bids, bsc, _ = eng.generate(*sub, num_beams=5, max_length=new + 1, min_length=new + 1)
    bids, bsc = eng.mem.numpy(bids).copy(), eng.mem.numpy(bsc).copy()
    assert bids.shape == g["beam_ids"].shape
    # sum of 16 log-probabilities / 16: the tolerance of one logit (log-softmax is 1-Lipschitz in the max norm, x2)
    np.testing.assert_allclose(bsc, g["beam_scores"], atol=2 * LOGIT_TOL)
    assert sum(np.array_equal(bids[b], g["beam_ids"][b]) for b in range(nb)) >= nb // 2      # measured: 3 of 4 rows identical
    ids32, sc32, _ = eng.generate(*args, num_beams=5, max_length=new + 1, min_length=new + 1)
    ids32, sc32 = eng.mem.numpy(ids32).copy(), eng.mem.numpy(sc32).copy()
    assert ids32.shape == (int(g["batch"]), new + 1) and np.all(ids32[:, 0] == 0) and ids32.min() >= 0 and ids32.max() < shape.vocab_size
    np.testing.assert_allclose(sc32[:nb], g["beam_scores"], atol=2 * LOGIT_TOL)
    assert sum(np.array_equal(ids32[b], g["beam_ids"][b]) for b in range(nb)) >= nb // 2
    again, sc_again, _ = eng.generate(*args, num_beams=5, max_length=new + 1, min_length=new + 1)
    assert np.array_equal(eng.mem.numpy(again), ids32) and np.array_equal(eng.mem.numpy(sc_again), sc32)


def test_g4_beam5_all_32_images_against_stock():
    """The reference's shipped decode mode (config/predict.yaml:13: beam search, 5 beams) on ALL 32 bench images against stock UDOP
    (tests/golden/g4_beam32.npz, tools/make_golden.py g4beam: best hypothesis, its score, the gap to stock's own second hypothesis).
    What can be asserted on a random-weight model: stock's best and second hypotheses are 0.0002 ... 0.012 apart in score (median
    0.0015) - far inside the bf16 tolerance of ONE logit - so which of the near-tied hypotheses wins is not decidable; the SCORE of the
    winner is: every image's sequence score within 2 x LOGIT_TOL of stock's best (a search that lost a beam, mis-ordered the top-2K
    or mis-applied the length penalty would sit far outside).  Ids: the share of images whose hypothesis is exactly stock's best or
    stock's second is reported and must not collapse.  Both the 160-row batch call and the beam queue (mg_generate_stream_beam)."""
    g, shape, eng, args = _setup()
    gb = load_golden("g4_beam32.npz")
    new = int(gb["new_tokens"])
    assert np.array_equal(gb["beam_ids"][:int(g["beam_rows"])], g["beam_ids"]) and float(gb["beam_gap"].max()) < 2 * LOGIT_TOL
    ids, sc, _ = eng.generate(*args, num_beams=5, max_length=new + 1, min_length=new + 1)
    ids, sc = eng.mem.numpy(ids).copy(), eng.mem.numpy(sc).copy()
    np.testing.assert_allclose(sc, gb["beam_scores"], atol=2 * LOGIT_TOL)
    hit = sum(bool(np.array_equal(ids[b], gb["beam_ids"][b]) or np.array_equal(ids[b], gb["beam_second_ids"][b])) for b in range(ids.shape[0]))
    print(f"beam-5, batch call: {hit} of {ids.shape[0]} hypotheses equal stock's best or second; max |score - stock| {np.abs(sc - gb['beam_scores']).max():.4f}")
    assert hit >= 23, hit              # measured 26 (threshold = measured - 3)
    qi, ql, qs, _ = eng.generate_stream_beam(*args, num_beams=5, max_length=new + 1, min_length=new + 1, chunk=16, slots=16, pool_chunks=3)
    qi, ql, qs = eng.mem.numpy(qi), eng.mem.numpy(ql), eng.mem.numpy(qs)
    assert np.all(ql == new + 1)
    np.testing.assert_allclose(qs, gb["beam_scores"], atol=2 * LOGIT_TOL)
    hitq = sum(bool(np.array_equal(qi[b], gb["beam_ids"][b]) or np.array_equal(qi[b], gb["beam_second_ids"][b])) for b in range(qi.shape[0]))
    print(f"beam-5, queue form: {hitq} of {qi.shape[0]} hypotheses equal stock's best or second")
    assert hitq >= 23, hitq            # measured 26 (threshold = measured - 3)


def test_g4_long_256_forced_steps_top8():
    """The benchmark's own decode length: 256 forced steps (bench.py: max_length = min_length = 257) on 4 of the bench images,
    teacher-forced along stock's ids through the KV-cached decode path; EVERY step's logits at stock's top-8 indices within
    LOGIT_TOL, ranks under the margin rule - positions 17..256 included (self-attention cache beyond one 128-key round, the
    log-bucketed and the saturated range of the decoder's positional bias).  Fixture: tests/golden/g4_long.npz (tools/make_golden.py
    g4long, stock UDOP; the oracle agrees with stock on the same 256 positions to `oracle_top8_maxdiff`)."""
    g, shape, eng0, args = _setup()
    gl = load_golden("g4_long.npz")
    rows, new = gl["rows"], int(gl["new_tokens"])
    assert new == 256 and np.array_equal(gl["greedy_ids"][:, :17], g["greedy_ids"][rows])
    sd = synth.recipe_state_dict(shape, **synth.BENCH_RECIPE)
    eng = make_engine("hip", shape, sd, max_decode_len=new + 1)
    sub = tuple(np.asarray(a[rows]) if isinstance(a, np.ndarray) else a[rows.tolist()] for a in args)
    cap = eng.debug_decode_capture(new, len(rows), gl["greedy_ids"])
    try:
        ids, _, _ = eng.generate(*sub, max_length=new + 1, min_length=new + 1)
        logits = eng.mem.numpy(cap).transpose(1, 0, 2).copy()          # [4, 256, V]
    finally:
        eng.debug_decode_capture()
    err = _check_top8(logits, gl["step_top_vals"], gl["step_top_idx"])
    # no drift with position: the last 128 steps are no worse than the first 64 (beyond noise)
    per_step = err.max(axis=(0, 2))
    assert per_step[128:].max() < max(1.5 * per_step[:64].max(), 0.5 * LOGIT_TOL), (per_step[:64].max(), per_step[128:].max())
    # free-running (what bench.py runs): ids equal stock's up to each row's first near-tie; report how far that is
    ids_free, _, _ = eng.generate(*sub, max_length=new + 1, min_length=new + 1)
    ids_free = eng.mem.numpy(ids_free)
    margin = gl["step_top_vals"][..., 0] - gl["step_top_vals"][..., 1]
    for b in range(len(rows)):
        for t in range(1, new + 1):
            if margin[b, t - 1] < MARGIN_TOL:
                break
            assert ids_free[b, t] == gl["greedy_ids"][b, t], (b, t)


def test_g4_four_batches_in_flight_equal_the_call_made_alone():
    """bench.py's own loop shape at production dimensions: the benchmark batch (B = 32, large shape) on four execution contexts at the
    same time, two rounds (capture, then replay): every context's ids and per-step top-2 logits are bit-identical to the one-at-a-time
    call - which the tests above pin on stock UDOP - and so is a different batch (images in reverse order) decoded beside them."""
    from markushgrapher_amd.inflight import InFlight
    g, shape, eng, args = _setup()
    new = int(g["new_tokens"])
    ids, bbox, mask, pix = args
    pix_np = eng.mem.numpy(pix).copy()
    batches = [(ids, bbox, mask, pix_np), (ids[::-1].copy(), bbox[::-1].copy(), mask[::-1].copy(), pix_np[::-1].copy())]

    def run(ctx, k):
        o, _, top2 = ctx.generate(*batches[k], max_length=new + 1, min_length=new + 1, return_top2=True)
        return ctx.mem.numpy(o).copy(), ctx.mem.numpy(top2).copy()
    want = [run(eng, 0), run(eng, 1)]
    assert np.array_equal(want[0][0], want[1][0][::-1])                  # rows do not depend on their place in the batch
    with InFlight(eng, 4) as fl:
        for _ in range(2):
            got = fl.map(run, [0, 1, 0, 0, 1, 0, 1, 1])
            for k, (o, t2) in zip([0, 1, 0, 0, 1, 0, 1, 1], got):
                assert np.array_equal(o, want[k][0]) and np.array_equal(t2, want[k][1])


def test_g4_with_the_ocsr_branch_attached_at_production_dimensions():
    """The reference's shipped architecture (me-lf-stack-1) at the benchmark's dimensions: UDOP-large + the Swin-B geometry attached
    (mg_attach_e1), B = 32 on the bench inputs.  (i) a call that lets the library evaluate the branch gives the same ids as the call that
    is handed the branch's tokens (same bits: the attached path is the precomputed-token path fed from the call's own workspace), for the
    32-row call and a 64-row call; (ii) 2 images against the CPU oracle chain SwinOracle.e1 -> Oracle(e1=...): the e1 block itself and
    teacher-forced logits at the reference's top-8 within LOGIT_TOL (fusion INFERRED: parity unpinned, oracle = the build's own);
    (iii) the tokens matter (logits differ from the VTL-only model's by far more than the tolerance)."""
    import torch
    from markushgrapher_amd.e1 import E1Engine
    from markushgrapher_amd import e1_shapes
    from oracle.swin_oracle import SwinOracle
    from oracle.udop_oracle import Oracle
    g, shape, eng, args = _setup()
    s1 = e1_shapes.PRESETS["swin_b_384"]
    sd1 = e1_shapes.recipe_state_dict(s1)
    e1e = E1Engine(s1).load_state_dict(sd1)
    new = int(g["new_tokens"])
    ids_plain, _, _ = eng.generate(*args, max_length=new + 1, min_length=new + 1)
    ids_plain = eng.mem.numpy(ids_plain).copy()
    eng.attach_e1(e1e)
    try:
        e1 = e1e.encode(args[3])
        own, _, _ = eng.generate(*args, max_length=new + 1, min_length=new + 1)
        pre, _, _ = eng.generate(*args, max_length=new + 1, min_length=new + 1, e1=e1)
        own, pre = eng.mem.numpy(own).copy(), eng.mem.numpy(pre).copy()
        assert np.array_equal(own, pre)
        assert (own != ids_plain).any()                      # 144 more keys per image change what is decoded
        two = tuple(torch.cat([a, a]) if torch.is_tensor(a) else np.concatenate([a, a]) for a in args)
        own64, _, _ = eng.generate(*two, max_length=new + 1, min_length=new + 1)
        own64 = eng.mem.numpy(own64)
        assert np.array_equal(own64[:32], own) and np.array_equal(own64[32:], own)       # rows do not depend on the call's row count
        # (ii) two images against the oracle chain
        nb, T = 2, 8
        sub = tuple(a[:nb] for a in args)
        labels = g["labels"][:nb, :T]
        dec_ids = Oracle.shift_right(labels, shape.decoder_start_token_id, shape.pad_token_id).numpy()
        dam = (labels != -100).astype(np.uint8)
        logits, _, _ = eng.forward_logits(*sub, dec_ids, dam)
        logits = eng.mem.numpy(logits)
        pix = eng.mem.numpy(args[3][:nb])
        with torch.no_grad():
            e1_ref = SwinOracle(s1, sd1).e1(pix).numpy()
        assert np.abs(eng.mem.numpy(e1[:nb]) - e1_ref).max() < 0.02 * np.abs(e1_ref).max() + 0.02
        sd = synth.recipe_state_dict(shape, **synth.BENCH_RECIPE)
        o = Oracle(shape, sd)
        inp = [np.asarray(a[:nb]) if isinstance(a, np.ndarray) else None for a in args[:3]]
        ref = o.forward(inp[0], inp[1], pix, inp[2], labels=labels, decoder_attention_mask=dam.astype(np.int64), e1=e1_ref).numpy()
        plain = o.forward(inp[0], inp[1], pix, inp[2], labels=labels, decoder_attention_mask=dam.astype(np.int64)).numpy()
        live = labels != -100
        top = np.argsort(-ref, axis=-1)[..., :8]
        err = np.abs(np.take_along_axis(logits, top, -1) - np.take_along_axis(ref, top, -1))[live]
        assert err.max() < LOGIT_TOL, err.max()
        assert np.abs(ref - plain)[live].max() > 5 * LOGIT_TOL
    finally:
        eng.attach_e1(None)
        e1e.close()
    back, _, _ = eng.generate(*args, max_length=new + 1, min_length=new + 1)
    assert np.array_equal(eng.mem.numpy(back), ids_plain)       # detached: the plain VTL model again, bit for bit


def test_g4_b1_second_pinned_batch_inside_a_160_row_call():
    """A SECOND batch of the benchmark's timed region pinned on stock UDOP: batch j = 1 of bench.py's pool (seed + 1000, padded to the pool's
    common text length; tests/golden/g4_bench_b1.npz, tools/make_golden.py g4b1), as the FIRST batch of a 160-row call whose five batches
    are all different from the batch g4_bench.npz pins (pool batches 1 .. 5: what the second context's call of the timed region holds).
    Encoder probes of the 32 images; free-running greedy ids under the margin rule with the per-step top-1 logit within LOGIT_TOL; and the
    call's first batch bit-identical to a call on that batch alone (the bench's own ids_equal_one_batch_calls check)."""
    g1 = load_golden("g4_bench_b1.npz")
    _, shape, eng, _ = _setup()
    B, new, L = int(g1["batch"]), int(g1["new_tokens"]), int(g1["text_len_padded"])
    assert int(g1["synth_seed"]) == synth.BENCH_SEED + 1000 and int(g1["pool_batch"]) == 1
    parts = []
    for j in range(1, 6):
        p = synth.synth_batch(shape, B, seed=synth.BENCH_SEED + 1000 * j, return_pages=True)
        padn = L - p["input_ids"].shape[1]
        assert padn >= 0
        if padn:
            p["input_ids"] = np.pad(p["input_ids"], ((0, 0), (0, padn)))
            p["attention_mask"] = np.pad(p["attention_mask"], ((0, 0), (0, padn)))
            p["bbox"] = np.pad(p["bbox"], ((0, 0), (0, padn), (0, 0)))
        parts.append(p)
    cat = {k: np.concatenate([p[k] for p in parts], 0) for k in parts[0]}
    pix = eng.preprocess(cat["pages_u8"])
    enc, mask = eng.encode(parts[0]["input_ids"], parts[0]["bbox"], parts[0]["attention_mask"], pix[:B])
    enc, mask = eng.mem.numpy(enc), eng.mem.numpy(mask)
    assert np.array_equal(mask, g1["enc_mask"].astype(np.uint8))
    for b in range(B):
        err = np.abs(enc[b][g1["enc_rows"][b]] - g1["enc_probe"][b])
        assert err.max() < ENC_MAX and err.mean() < ENC_MEAN, (b, err.max(), err.mean())
    ids, _, top2 = eng.generate(cat["input_ids"], cat["bbox"], cat["attention_mask"], pix, max_length=new + 1, min_length=new + 1, return_top2=True)
    ids, top2 = eng.mem.numpy(ids).copy(), eng.mem.numpy(top2).copy()
    assert ids.shape == (5 * B, new + 1)
    ref, vals = g1["greedy_ids"], g1["step_top_vals"]
    margin = vals[..., 0] - vals[..., 1]
    compared = 0
    for b in range(B):
        for t in range(1, new + 1):
            if margin[b, t - 1] < MARGIN_TOL:
                break
            assert ids[b, t] == ref[b, t], (b, t)
            assert abs(top2[t, b, 0] - vals[b, t - 1, 0]) < LOGIT_TOL
            compared += 1
    expected = sum(int(np.argmax(np.append(margin[b] < MARGIN_TOL, True))) for b in range(B))
    same_rows = int((ids[:B] == ref).all(1).sum())
    print(f"G4-b1 inside a 160-row call: {compared} positions compared under the margin rule, {same_rows} of 32 rows identical to stock over all 16 steps")
    assert compared == expected and compared >= 30, (compared, expected)
    one, _, _ = eng.generate(parts[0]["input_ids"], parts[0]["bbox"], parts[0]["attention_mask"], pix[:B], max_length=new + 1, min_length=new + 1)
    assert np.array_equal(eng.mem.numpy(one), ids[:B])


def test_g4_greedy_kv_form_under_margin_rule():
    """The K / V form of the greedy cross-attention (what calls below 96 decode rows take by default, mg_set_cross_absorb) on the same
    fixture: ids under the margin rule and the per-step top-1 logit within LOGIT_TOL."""
    g, shape, eng, args = _setup()
    new = int(g["new_tokens"])
    eng.set_cross_absorb(False)
    try:
        ids, _, top2 = eng.generate(*args, max_length=new + 1, min_length=new + 1, return_top2=True)
        ids, top2 = eng.mem.numpy(ids).copy(), eng.mem.numpy(top2).copy()
    finally:
        eng.set_cross_absorb(True)
    ref, vals = g["greedy_ids"], g["step_top_vals"]
    margin = vals[..., 0] - vals[..., 1]
    compared = 0
    for b in range(ref.shape[0]):
        for t in range(1, new + 1):
            if margin[b, t - 1] < MARGIN_TOL:
                break
            assert ids[b, t] == ref[b, t], (b, t)
            assert abs(top2[t, b, 0] - vals[b, t - 1, 0]) < LOGIT_TOL
            compared += 1
    assert compared >= 50
