"""CPU: the oracle (oracle/udop_oracle.py) against the golden vectors minted from stock transformers' UDOP
(tools/make_golden.py).  This is what pins the oracle; the HIP path is then compared with the oracle
(tests/test_gpu_*.py) and with the same golden vectors."""
import numpy as np
import pytest

from markushgrapher_amd import synth
from oracle.udop_oracle import Oracle, bucket_table
from tests.conftest import load_golden, GOLDEN
import os


def _weights(g):
    shape = synth.SHAPES[str(g["shape"])]
    if "recipe" in g:
        r = dict(zip(("gain", "embed_gain", "ffn_gain", "xq_gain"), (float(x) for x in g["recipe"])))
        return shape, synth.recipe_state_dict(shape, **r)
    if "gain" in g:
        return shape, synth.recipe_state_dict(shape, gain=float(g["gain"]))
    return shape, dict(np.load(os.path.join(GOLDEN, "g3_weights.npz")))


def _inputs(g, shape):
    if "input_ids" in g:
        return {k: g[k] for k in ("input_ids", "bbox", "attention_mask", "pixel_values")}
    B, lo, hi, seed = [int(x) for x in g["synth_args"]]
    return synth.synth_batch(shape, B, L_min=lo, L_max=hi, seed=seed)


@pytest.mark.parametrize("name", ["g0_tiny.npz", "g3_trained_tiny.npz", "g1_mid.npz"])
def test_oracle_matches_stock_fixture(name):
    g = load_golden(name)
    shape, sd = _weights(g)
    inp = _inputs(g, shape)
    o = Oracle(shape, sd)
    enc, mask = o.encode(inp["input_ids"], inp["bbox"], inp["pixel_values"], inp["attention_mask"])
    assert np.array_equal(mask.numpy(), g["enc_mask"])
    valid = g["enc_mask"].astype(bool)
    assert np.abs(enc.numpy() - g["enc_out"])[valid].max() < 2e-4
    enc2, _ = o.encode(inp["input_ids"], inp["bbox"], inp["pixel_values"], None)
    assert np.abs(enc2.numpy() - g["enc_out_nomask"]).max() < 2e-4
    labels = g["labels"]
    logits = o.forward(inp["input_ids"], inp["bbox"], inp["pixel_values"], inp["attention_mask"], labels=labels,
                       decoder_attention_mask=(labels != -100).astype(np.int64))
    assert np.abs(logits.numpy() - g["logits"]).max() < 1e-3
    ml = int(g["max_length"])
    ids = o.greedy(inp["input_ids"], inp["bbox"], inp["pixel_values"], inp["attention_mask"], max_length=ml)
    assert np.array_equal(ids, g["greedy_ids"])
    bids, bsc = o.beam_search(inp["input_ids"], inp["bbox"], inp["pixel_values"], inp["attention_mask"],
                              num_beams=5, max_length=ml)
    assert np.array_equal(bids, g["beam_ids"])
    assert np.abs(bsc - g["beam_scores"]).max() < 1e-3


@pytest.mark.parametrize("name", ["g0_tiny.npz", "g1_mid.npz"])
def test_random_weight_fixtures_are_not_degenerate(name):
    """G0 / G1 (recipe weights): greedy and beam rows are varied, image-dependent sequences - not the all-start-token rows that
    plain random init gives, on which an id comparison says nothing."""
    g = load_golden(name)
    for ids, distinct in ((g["greedy_ids"], 5), (g["beam_ids"], 2)):
        assert (ids[:, 1:] != 0).all() and all(len(set(r[1:].tolist())) >= distinct for r in ids)
        assert len({tuple(r.tolist()) for r in ids}) == ids.shape[0]


def test_g3_has_early_eos_and_margins():
    g = load_golden("g3_trained_tiny.npz")
    ids = g["greedy_ids"]
    eos_pos = [int(np.argmax(r == 1)) for r in ids]
    assert len(set(eos_pos)) >= 3, "rows must emit EOS at different steps"
    live = np.zeros_like(g["greedy_margin"], dtype=bool)
    for b, p in enumerate(eos_pos):
        live[b, :p] = True                 # steps that produced tokens 1..eos
    assert g["greedy_margin"][live].min() > 0.4


def test_bucket_tables_sane():
    t = load_golden("bucket_tables.npz")
    assert np.array_equal(t["enc_1d"], bucket_table(True, 32, 128, -300, 300))
    lo = int(t["enc_1d_lo"])
    # exact power-of-two boundaries of the encoder 1-D table
    for n, b in ((16, 10), (32, 12), (64, 14), (128, 15)):
        assert t["enc_1d"][n - lo] == 16 + b and t["enc_1d"][-n - lo] == b
