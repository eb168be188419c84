"""Several batches in flight (C ABI mg_clone; markushgrapher_amd/inflight.py): further execution contexts on one set of weights,
each with its own workspace / decode graph / stream / host thread.  What must hold: a context's results are exactly those of the
context it was cloned from, whether the calls run one after the other (`emu`: the CPU SIMT emulator is single-threaded test
infrastructure) or overlap in time on the GPU (`hip`), greedy and beam, and weights loaded later through the source are seen by
every context.  The reference has no counterpart: it runs one batch at a time (utils_evaluation.py:269-285)."""
import numpy as np
import pytest

from markushgrapher_amd import synth
from markushgrapher_amd.inflight import InFlight
from tests.backends import make_engine
from tests.conftest import load_golden
from tests.test_oracle_golden import _weights, _inputs

BACKENDS = [pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)]


def _gen(eng, inp, **kw):
    ids, _, _ = eng.generate(inp["input_ids"], inp["bbox"], inp["attention_mask"], inp["pixel_values"], **kw)
    return eng.mem.numpy(ids).copy()


@pytest.mark.parametrize("be_name", BACKENDS)
def test_clone_matches_source_and_golden(be_name):
    g = load_golden("g3_trained_tiny.npz")
    shape, sd = _weights(g)
    inp = _inputs(g, shape)
    T = int(g["max_length"])
    eng = make_engine(be_name, shape, sd)
    ctx = eng.clone()
    a = _gen(eng, inp, max_length=T)
    b = _gen(ctx, inp, max_length=T)
    assert np.array_equal(a, b)
    assert np.array_equal(b, g["greedy_ids"][:, :b.shape[1]])
    # beam search on the clone, greedy on the source in between (each context has its own captured step)
    bb = _gen(ctx, inp, max_length=T, num_beams=5)
    assert np.array_equal(bb, g["beam_ids"][:, :bb.shape[1]])
    if be_name == "hip":                      # (the emulator takes 10 s per beam call)
        a2 = _gen(eng, inp, max_length=T)
        ba = _gen(eng, inp, max_length=T, num_beams=5)
        assert np.array_equal(a2, a) and np.array_equal(ba, bb)
    # encoder output through the clone
    ea, _ = eng.encode(inp["input_ids"], inp["bbox"], inp["attention_mask"], inp["pixel_values"])
    eb, _ = ctx.encode(inp["input_ids"], inp["bbox"], inp["attention_mask"], inp["pixel_values"])
    assert np.array_equal(eng.mem.numpy(ea), ctx.mem.numpy(eb))
    ctx.close()
    assert np.array_equal(_gen(eng, inp, max_length=6), a[:, :6])   # the source outlives its clones


@pytest.mark.parametrize("be_name", BACKENDS)
def test_clone_sees_later_weights(be_name):
    g = load_golden("g3_trained_tiny.npz")
    shape, sd = _weights(g)
    inp = _inputs(g, shape)
    eng = make_engine(be_name, shape, sd)
    ctx = eng.clone()
    T = 10
    base = _gen(ctx, inp, max_length=T, min_length=T)
    emb = np.array(sd["shared.weight"], dtype=np.float32).copy()
    emb[5] *= 6.0                                                       # makes token 5 the arg-max of most steps
    eng.load_state_dict({"shared.weight": emb})
    a = _gen(eng, inp, max_length=T, min_length=T)
    b = _gen(ctx, inp, max_length=T, min_length=T)
    assert np.array_equal(a, b) and not np.array_equal(b, base)


def test_clone_refuses_unfinalized():
    g = load_golden("g3_trained_tiny.npz")
    shape, sd = _weights(g)
    eng = make_engine("emu", shape, None)
    with pytest.raises(Exception, match="finalized"):
        eng.clone()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4])
def test_batches_in_flight_equal_serial_calls(n):
    """n worker threads, one context + stream each, overlapping on the GPU; different inputs per job so a mix-up would show."""
    import torch
    from markushgrapher_amd.inflight import InFlight
    g = load_golden("g3_trained_tiny.npz")
    shape, sd = _weights(g)
    inp = _inputs(g, shape)
    B = inp["input_ids"].shape[0]
    T = 24
    eng = make_engine("hip", shape, sd)
    jobs = []
    rng = np.random.default_rng(5)
    for j in range(4 * n + 1):
        order = rng.permutation(B)[: 2 + j % (B - 1)]
        jobs.append({k: np.ascontiguousarray(v[order]) for k, v in inp.items()})
    want = [_gen(eng, jb, max_length=T, min_length=T) for jb in jobs]
    want_beam = [_gen(eng, jb, max_length=T, num_beams=3) for jb in jobs[:n + 1]]

    def greedy(ctx, jb):
        return _gen(ctx, jb, max_length=T, min_length=T)

    def beam(ctx, jb):
        return _gen(ctx, jb, max_length=T, num_beams=3)

    with InFlight(eng, n) as fl:
        assert len(fl) == n
        for _ in range(2):                                   # second pass: replayed graphs on every context
            got = fl.map(greedy, jobs)
            for w, o in zip(want, got):
                assert np.array_equal(w, o)
        gb = fl.map(beam, jobs[:n + 1])
        for w, o in zip(want_beam, gb):
            assert np.array_equal(w, o)
        # mixed: greedy and beam calls overlapping
        futs = [fl.submit(beam if i % 2 else greedy, jobs[i % (n + 1)]) for i in range(2 * n)]
        for i, f in enumerate(futs):
            assert np.array_equal(f.result(), (want_beam if i % 2 else want)[i % (n + 1)])
    torch.cuda.synchronize()
    assert np.array_equal(_gen(eng, jobs[0], max_length=T, min_length=T), want[0])


def test_inflight_helper_on_emulator_keeps_order_and_contexts():
    """Host logic of InFlight without a GPU: jobs are spread over the contexts, results come back in submission order."""
    from markushgrapher_amd.inflight import InFlight
    g = load_golden("g3_trained_tiny.npz")
    shape, sd = _weights(g)
    inp = _inputs(g, shape)
    eng = make_engine("emu", shape, sd)
    jobs = [{k: np.ascontiguousarray(v[i:i + 2]) for k, v in inp.items()} for i in range(4)]
    want = [_gen(eng, jb, max_length=8, min_length=8) for jb in jobs]
    seen = set()

    def fn(ctx, jb):
        seen.add(id(ctx))
        return _gen(ctx, jb, max_length=8, min_length=8)

    with InFlight(eng, 2) as fl:
        got = fl.map(fn, jobs)
    assert all(np.array_equal(w, o) for w, o in zip(want, got))
    assert 1 <= len(seen) <= 2
    with pytest.raises(ValueError):
        InFlight(eng, 5)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_shared_gpu_setting_changes_no_bits(be_name):
    """mg_set_shared_gpu (one resident workgroup of the cross-attention stream per CU while other contexts run beside this one) is a
    scheduling hint: ids with it on equal ids with it off; InFlight switches it on for its contexts and gives the source engine its
    previous setting back."""
    g = load_golden("g3_trained_tiny.npz")
    shape, sd = _weights(g)
    inp = _inputs(g, shape)
    eng = make_engine(be_name, shape, sd)
    off = _gen(eng, inp, max_length=10, min_length=10)
    assert eng.set_shared_gpu(True) is False
    on = _gen(eng, inp, max_length=10, min_length=10)
    assert np.array_equal(on, off)
    assert eng.set_shared_gpu(False) is True
    with InFlight(eng, 2) as fl:
        assert all(c.set_shared_gpu(True) is True for c in fl.contexts)          # InFlight had switched it on
        got = fl.map(lambda ctx, _: _gen(ctx, inp, max_length=10, min_length=10), range(2))
    assert all(np.array_equal(o, off) for o in got)
    assert eng.set_shared_gpu(False) is False                                     # restored by close()


def test_plan_calls_keeps_every_context_busy():
    from markushgrapher_amd.inflight import plan_calls
    assert plan_calls(20, 4, 4) == [3, 3, 3, 3, 2, 2, 2, 2]
    assert plan_calls(16, 4, 4) == [4, 4, 4, 4]
    assert plan_calls(8, 4, 4) == [2, 2, 2, 2]
    assert plan_calls(5, 4, 4) == [2, 1, 1, 1]
    assert plan_calls(3, 4, 4) == [1, 1, 1]
    assert plan_calls(0, 4, 4) == []
    assert plan_calls(7, 1, 3) == [3, 2, 2]
    for k in range(0, 40):
        for n in (1, 2, 4):
            for m in (1, 2, 4):
                p = plan_calls(k, n, m)
                assert sum(p) == k and all(1 <= x <= m for x in p)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_generate_batches_equals_one_call_per_batch(be_name):
    """InFlight.generate_batches (several batches per call, calls spread over the contexts) returns for every batch the ids of a call
    on the batch alone (two contexts, five batches of 9 rows: calls of 2 and 1 batches; on the GPU the calls overlap in time)."""
    shape = synth.SHAPES["tiny" if be_name == "emu" else "mid"]
    sd = synth.recipe_state_dict(shape, **synth.BENCH_RECIPE)
    eng = make_engine(be_name, shape, sd, max_decode_len=32)
    inp = synth.synth_batch(shape, 45, L_min=10, L_max=16, seed=3)
    batches = [{k: v[9 * i:9 * i + 9] for k, v in inp.items()} for i in range(5)]
    ref = [_gen(eng, b, max_length=8, min_length=8) for b in batches]
    with InFlight(eng, 2) as fl:
        got = fl.generate_batches(batches, max_batches_per_call=2, max_length=8, min_length=8)
    assert len(got) == 5
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)


def test_one_call_at_a_time_per_context():
    """A context refuses a second call while it is inside one (two host threads on ONE context): MG_E_STATE, the first call's result intact.
    (The emulator backend: a generate call takes seconds and ctypes releases the GIL, so the overlap is certain.)"""
    import threading
    import time
    from markushgrapher_amd.engine import MgError
    g = load_golden("g3_trained_tiny.npz")
    shape, sd = _weights(g)
    inp = _inputs(g, shape)
    eng = make_engine("emu", shape, sd)
    want = _gen(eng, inp, max_length=10, min_length=10)
    out = {}

    def first():
        out["ids"] = _gen(eng, inp, max_length=10, min_length=10)
    t = threading.Thread(target=first)
    t.start()
    time.sleep(0.3)
    with pytest.raises(MgError, match="one call at a time"):
        eng.encode(inp["input_ids"], inp["bbox"], inp["attention_mask"], inp["pixel_values"])
    t.join()
    assert np.array_equal(out["ids"], want)
    assert np.array_equal(_gen(eng, inp, max_length=10, min_length=10), want)       # and the context is usable afterwards
