"""Continuous decoding (C ABI mg_generate_stream): `slots` decode rows work through a queue of images, a row that ends hands its
slot to the next image, the encoder of the next chunk runs ahead of the decode steps.  What must hold (and is all that is new):
every image's ids are EXACTLY the ids mg_generate returns for that image - whatever slot it sat in, whatever its neighbours were,
whenever it started - and the lengths are the per-image lengths HF's generate(max_length) would return at batch size 1, which is
how the reference calls it (ref: utils/ocsr/utils_evaluation.py:140, 269-285).  `emu` = the same sources on the CPU SIMT emulator
(synchronous: the encoder 'stream' is the caller's); `hip` = MI355X with the encoder on its own stream."""
import numpy as np
import pytest

from markushgrapher_amd import synth
from tests.backends import make_engine
from tests.conftest import load_golden
from tests.test_oracle_golden import _weights, _inputs

BACKENDS = [pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)]


def _np(eng, h):
    return eng.mem.numpy(h)


def _queue(inp, order):
    return {k: np.ascontiguousarray(v[order]) for k, v in inp.items()}


def _batch_rows(eng, inp, max_length, min_length=0):
    """generate() per image, one at a time (batch size 1, as the reference) -> list of id rows."""
    rows = []
    for b in range(inp["input_ids"].shape[0]):
        one = {k: v[b:b + 1] for k, v in inp.items()}
        ids, _, _ = eng.generate(one["input_ids"], one["bbox"], one["attention_mask"], one["pixel_values"], max_length=max_length,
                                 min_length=min_length)
        rows.append(_np(eng, ids)[0].copy())
    return rows


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("chunk,slots,pool_chunks,absorb", [(4, 3, 2, False), (3, 5, 3, False), (6, 2, 2, False), (3, 5, 3, True), (4, 3, 2, True)])
def test_stream_ids_equal_per_image_generate(be_name, chunk, slots, pool_chunks, absorb):
    """absorb: the cross-attention form pinned (mg_set_cross_absorb; the default picks it by the decode rows, i.e. by `slots`) - the
    weight-absorbed form reads the image's attended encoder states from the pool through the slot table, skips idle slots and pads the
    last 16-key stage of every pool entry."""
    g = load_golden("g3_trained_tiny.npz")
    shape, sd = _weights(g)
    inp = _inputs(g, shape)                                    # 6 images whose rows end at different steps
    order = np.array([0, 3, 5, 1, 2, 4, 4, 0, 1, 5, 2, 3, 3, 0])      # 14 images in the queue (not a multiple of any chunk)
    q = _queue(inp, order)
    T = int(g["max_length"])
    eng = make_engine(be_name, shape, sd)
    eng.set_cross_absorb(absorb)
    ids, lens, steps = eng.generate_stream(q["input_ids"], q["bbox"], q["attention_mask"], q["pixel_values"], max_length=T,
                                           chunk=chunk, slots=slots, pool_chunks=pool_chunks)
    ids, lens = _np(eng, ids), _np(eng, lens)
    ref = g["greedy_ids"]
    for n, b in enumerate(order):
        row = ref[b]
        e = np.nonzero(row == shape.eos_token_id)[0]
        want_len = int(e[0]) + 1 if len(e) else T
        assert lens[n] == want_len, (n, b, lens[n], want_len)
        assert np.array_equal(ids[n, :want_len], row[:want_len]), (n, b)
        assert np.all(ids[n, want_len:] == shape.pad_token_id)
    # no step is spent waiting for a batch's longest member: the total is (close to) sum of lengths / slots, not chunks x longest
    total_tokens = int(sum(l - 1 for l in lens))
    assert steps >= -(-total_tokens // slots)
    assert steps <= total_tokens // slots + len(order) + 48, (steps, total_tokens)     # slack: chunk hand-over + the host's late view at the end


@pytest.mark.parametrize("be_name", BACKENDS)
def test_stream_forced_length_and_small_queues(be_name):
    """EOS suppressed (min_length = max_length, the headline benchmark's setting): all rows run to max_length; and queues smaller than
    the slot count / a single image."""
    g = load_golden("g3_trained_tiny.npz")
    shape, sd = _weights(g)
    inp = _inputs(g, shape)
    eng = make_engine(be_name, shape, sd)
    T = 12
    want = _batch_rows(eng, inp, T, T)
    order = np.array([2, 0, 5, 1, 4, 3, 2])
    q = _queue(inp, order)
    ids, lens, steps = eng.generate_stream(q["input_ids"], q["bbox"], q["attention_mask"], q["pixel_values"], max_length=T, min_length=T,
                                           chunk=4, slots=4, pool_chunks=2)
    ids, lens = _np(eng, ids), _np(eng, lens)
    assert np.all(lens == T) and not np.any(ids[:, 1:] == shape.eos_token_id)
    for n, b in enumerate(order):
        assert np.array_equal(ids[n], want[b]), (n, b)
    for N, slots in ((1, 4), (2, 8)):
        sub = _queue(inp, order[:N])
        ids2, lens2, _ = eng.generate_stream(sub["input_ids"], sub["bbox"], sub["attention_mask"], sub["pixel_values"], max_length=T, min_length=T,
                                             chunk=4, slots=slots, pool_chunks=2)
        assert np.array_equal(_np(eng, ids2), ids[:N]) and np.all(_np(eng, lens2) == T)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_stream_rejects_bad_arguments(be_name):
    from markushgrapher_amd.engine import MgError
    g = load_golden("g3_trained_tiny.npz")
    shape, sd = _weights(g)
    inp = _inputs(g, shape)
    eng = make_engine(be_name, shape, sd)
    args = (inp["input_ids"], inp["bbox"], inp["attention_mask"], inp["pixel_values"])
    with pytest.raises(MgError, match="max_length"):
        eng.generate_stream(*args, max_length=100000)
    with pytest.raises(MgError, match="pool"):
        eng.generate_stream(*args, max_length=8, chunk=2, slots=8, pool_chunks=2)
    with pytest.raises(MgError, match="slots"):
        eng.generate_stream(*args, max_length=8, chunk=300, slots=300, pool_chunks=2)
    bad = inp["input_ids"].copy()
    bad[4, 0] = shape.vocab_size + 1
    with pytest.raises(MgError, match="token ids"):
        eng.generate_stream(bad, *args[1:], max_length=8, chunk=2, slots=2, pool_chunks=2)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_stream_large_shape_matches_batch_generate(mode):
    """UDOP-large shape, the bench recipe: 40 images through 16 slots (chunk 16) with EOS live, against mg_generate of the same images
    in batches; encoder on the caller's stream (0), on its own low-priority stream (1) and on a CU-masked stream (2) - the three
    placements must give identical ids (the overlap changes timing only)."""
    shape = synth.SHAPES["large"]
    sd = synth.recipe_state_dict(shape, **synth.BENCH_RECIPE)
    emb = sd["shared.weight"].copy()
    emb[shape.eos_token_id] = synth.round_bf16(emb[shape.eos_token_id] * np.float32(12.0))     # rows end at different steps (bench.py's EOS run)
    sd["shared.weight"] = emb
    eng = make_engine("hip", shape, sd, max_decode_len=64)
    inp = synth.synth_batch(shape, 40, seed=77, L_min=40, L_max=120)
    T = 48
    want, wlen = [], []
    for c0 in range(0, 40, 8):
        sl = {k: v[c0:c0 + 8] for k, v in inp.items()}
        ids, _, _ = eng.generate(sl["input_ids"], sl["bbox"], sl["attention_mask"], sl["pixel_values"], max_length=T)
        ids = _np(eng, ids)
        for r in ids:
            e = np.nonzero(r == shape.eos_token_id)[0]
            n = int(e[0]) + 1 if len(e) else T
            want.append(np.concatenate([r[:n], np.full(T - n, shape.pad_token_id, r.dtype)]) if len(r) >= n else None)
            wlen.append(n)
    eng.set_stream_encoder(mode, cu_mask=range(0, 256, 4) if mode == 2 else None)
    ids, lens, steps = eng.generate_stream(inp["input_ids"], inp["bbox"], inp["attention_mask"], inp["pixel_values"], max_length=T,
                                           chunk=16, slots=16, pool_chunks=2)
    ids, lens = _np(eng, ids), _np(eng, lens)
    assert len(set(wlen)) > 3, wlen                       # the workload really is ragged
    for n in range(40):
        assert lens[n] == wlen[n], (n, lens[n], wlen[n])
        assert np.array_equal(ids[n, :wlen[n]], want[n][:wlen[n]]), n
    eng.set_stream_encoder(1)


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("chunk,slots,pool_chunks,early", [(4, 3, 2, False), (3, 2, 3, True), (6, 5, 2, False)])
def test_beam_stream_equals_per_image_beam_search(be_name, chunk, slots, pool_chunks, early):
    """mg_generate_stream_beam: image slots of K = 3 beams work through a queue of 11 images (the trained fixture's 6, repeated in another
    order: their searches stop at different steps).  Every image's best hypothesis, its length and its score equal what
    generate(num_beams=3) returns for that image alone - the reference's call (batch size 1, utils_evaluation.py:140, 269-285)."""
    g = load_golden("g3_trained_tiny.npz")
    shape, sd = _weights(g)
    inp = _inputs(g, shape)
    K, T = 3, int(g["max_length"])
    eng = make_engine(be_name, shape, sd)
    want = []
    for b in range(inp["input_ids"].shape[0]):
        one = {k: v[b:b + 1] for k, v in inp.items()}
        ids, scores, _ = eng.generate(one["input_ids"], one["bbox"], one["attention_mask"], one["pixel_values"], num_beams=K, max_length=T,
                                      early_stopping=early)
        ids = _np(eng, ids)
        want.append((ids[0].copy(), float(_np(eng, scores)[0]), int(ids.shape[1])))
    order = np.array([0, 3, 5, 1, 2, 4, 4, 0, 1, 5, 2])
    q = _queue(inp, order)
    ids, lens, scores, steps = eng.generate_stream_beam(q["input_ids"], q["bbox"], q["attention_mask"], q["pixel_values"], num_beams=K,
                                                         max_length=T, early_stopping=early, chunk=chunk, slots=slots, pool_chunks=pool_chunks)
    ids, lens, scores = _np(eng, ids), _np(eng, lens), _np(eng, scores)
    for n, b in enumerate(order):
        row, sc, cols = want[b]
        assert lens[n] == cols, (n, b, lens[n], cols)
        assert np.array_equal(ids[n, :cols], row[:cols]), (n, b, ids[n].tolist(), row.tolist())
        assert scores[n] == np.float32(sc), (n, b, scores[n], sc)
    assert steps <= int(sum(lens)) // slots + len(order) + 48
