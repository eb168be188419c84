"""OCSR vision branch "e1" (SURVEY.md §8 rows a7 / f-2): oracle vs the stock-SwinModel golden vectors, checkpoint key maps, and the
C ABI `mg_e1_*` against both.

Fixtures (tools/make_golden_swin.py, minted from stock transformers SwinModel on recipe weights and inputs):
  swin_tiny.npz    window 4, three stages (16 / 8 / 4 maps: shifted windows with the region mask, two merges, a last stage whose
                   window is the whole map), 64 px - every code path at a size the emulator finishes in seconds
  swin_w12.npz     window 12 (the reference's) at 96 px: 24 x 24 and 12 x 12 maps
  swin_b_384.npz   Swin-B geometry (MolScribe's swin_base_patch4_window12_384: 86.88 M parameters, [B, 144, 1024]); probe rows +
                   per-row statistics of stock's output

What is pinned and what is not: `features` (SwinModel.last_hidden_state) is stock's output - PINNED.  The resize in front
(`derive_input`) and the projector behind (`e1`) restate INFERRED pieces of the reference's fork (e1_shapes.py): the HIP path is
compared with the build's own oracle there - parity unpinned, stated in DESIGN.md.

Tolerances: the HIP path keeps weights and GEMM / attention operands in bf16 with fp32 accumulation, an fp32 residual stream and
fp32 LayerNorm statistics; the golden vectors are fp32.  The features are LayerNorm outputs (|x| mean 0.8, max < 5):
  * vs stock fp32:               max-abs < 0.08, mean-abs < 0.012     (the bf16-emulating oracle itself sits at 0.015 / 0.002)
  * vs the bf16-emulating oracle: max-abs < 0.03                      (same storage points, different summation order)"""
import dataclasses

import numpy as np
import pytest

from markushgrapher_amd import e1_shapes
from markushgrapher_amd.e1_shapes import PRESETS, recipe_state_dict, synth_pixels, state_dict_spec
from tests.backends import get_backend, NumpyMem
from tests.conftest import load_golden

BACKENDS = [pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)]
FEAT_MAX, FEAT_MEAN, EMU_MAX = 0.08, 0.012, 0.03


def make_e1(be_name, s, sd):
    from markushgrapher_amd.e1 import E1Engine
    be = get_backend(be_name)
    eng = E1Engine(s, lib=be.lib, mem=NumpyMem()) if be_name == "emu" else E1Engine(s)
    return eng.load_state_dict(sd)


def _np(x):
    return x if isinstance(x, np.ndarray) else x.detach().cpu().numpy()


# ---- CPU: oracle and host logic -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny", "w12"])
def test_oracle_reproduces_stock(name):
    import torch
    from oracle.swin_oracle import SwinOracle
    g = load_golden(f"swin_{name}.npz")
    s = PRESETS[name]
    sd = recipe_state_dict(s)
    orc = SwinOracle(s, sd)
    with torch.no_grad():
        pix = orc.derive_input(synth_pixels(s, int(g["B"])))
        assert np.allclose(pix.double().sum(dim=(1, 2, 3)).numpy(), g["input_checksum"], rtol=1e-6)
        f = orc.features(pix).numpy()
    assert f.shape == g["features"].shape == (int(g["B"]), s.out_tokens, s.out_dim)
    assert np.abs(f - g["features"]).max() < 2e-4


def test_oracle_reproduces_stock_swin_b_probes():
    """Swin-B geometry (86.88 M parameters): the oracle against stock's probe rows and per-row statistics."""
    import torch
    from oracle.swin_oracle import SwinOracle
    g = load_golden("swin_b_384.npz")
    s = PRESETS["swin_b_384"]
    assert int(g["n_params"]) == 86878584 and s.out_tokens == 144 and s.out_dim == 1024
    sd = recipe_state_dict(s)
    assert sum(int(np.prod(v.shape)) for k, v in sd.items() if k.startswith("swin.")) == int(g["n_params"])
    orc = SwinOracle(s, sd)
    with torch.no_grad():
        f = orc.features(orc.derive_input(synth_pixels(s, 1))).numpy()
    assert np.abs(f[0, g["probe_rows"]] - g["features_probe"][0]).max() < 2e-4
    assert np.abs(f[0].mean(-1) - g["features_row_mean"][0]).max() < 1e-5
    assert np.abs(np.abs(f[0]).mean(-1) - g["features_row_absmean"][0]).max() < 1e-4


def _to_timm(sd, prefix="cnn."):
    """canonical -> timm 0.4.12 SwinTransformer names (fused qkv, buffers, a head), as MolScribe's checkpoint has them"""
    out = {}
    blocks = {}
    for k, v in sd.items():
        if not k.startswith("swin."):
            continue
        r = k[5:]
        r = r.replace("embeddings.patch_embeddings.projection.", "patch_embed.proj.").replace("embeddings.norm.", "patch_embed.norm.")
        if r.startswith("layernorm."):
            out[prefix + "norm." + r[len("layernorm."):]] = v
            continue
        r = r.replace("encoder.layers.", "layers.")
        r = r.replace("layernorm_before.", "norm1.").replace("layernorm_after.", "norm2.").replace("attention.o_proj.", "attn.proj.")
        r = r.replace("attention.relative_position_bias.relative_position_bias_table", "attn.relative_position_bias_table")
        hit = next((n for n in ("q_proj", "k_proj", "v_proj") if f"attention.{n}." in r), None)
        if hit:
            stem, leaf = r.split(f"attention.{hit}.")
            blocks.setdefault((stem, leaf), {})[hit] = v
            continue
        out[prefix + r] = v
    for (stem, leaf), parts in blocks.items():
        out[prefix + stem + "attn.qkv." + leaf] = np.concatenate([parts["q_proj"], parts["k_proj"], parts["v_proj"]], axis=0)
        out[prefix + stem + "attn.relative_position_index"] = np.zeros((4, 4), np.int64)
    out[prefix + "layers.0.blocks.1.attn_mask"] = np.zeros((4, 16, 16), np.float32)
    out[prefix + "head.weight"] = np.zeros((10, 8), np.float32)
    return out


def test_checkpoint_key_maps_round_trip():
    """timm (MolScribe), transformers-4.x and transformers-5.x Swin state dicts map onto the canonical keys with identical values; the
    projector's Linear layers are found in an nn.Sequential's numbering."""
    s = PRESETS["tiny"]
    sd = recipe_state_dict(s)
    want = {k: v for k, v in sd.items() if k.startswith("swin.")}
    got = e1_shapes.canonical_encoder_keys(_to_timm(sd))
    assert set(got) == set(want)
    assert all(np.array_equal(got[k], want[k]) for k in want)
    hf5 = {"encoder.molscribe." + k[5:]: v for k, v in want.items()}
    got = e1_shapes.canonical_encoder_keys(hf5)
    assert set(got) == set(want) and all(np.array_equal(got[k], want[k]) for k in want)
    hf4 = {}
    for k, v in want.items():
        r = k[5:]
        r = r.replace("attention.q_proj.", "attention.self.query.").replace("attention.k_proj.", "attention.self.key.")
        r = r.replace("attention.v_proj.", "attention.self.value.").replace("attention.o_proj.", "attention.output.dense.")
        r = r.replace("attention.relative_position_bias.relative_position_bias_table", "attention.self.relative_position_bias_table")
        r = r.replace("mlp.fc1.", "intermediate.dense.").replace("mlp.fc2.", "output.dense.")
        hf4[r] = v
    hf4["encoder.layers.0.blocks.0.attention.self.relative_position_index"] = np.zeros((16, 16), np.int64)
    got = e1_shapes.canonical_encoder_keys(hf4)
    assert set(got) == set(want) and all(np.array_equal(got[k], want[k]) for k in want)
    seq = {"0.weight": sd["proj.0.weight"], "0.bias": sd["proj.0.bias"], "2.weight": sd["proj.1.weight"], "2.bias": sd["proj.1.bias"]}
    pj = e1_shapes.canonical_projector_keys(seq)
    assert all(np.array_equal(pj[k], sd[k]) for k in ("proj.0.weight", "proj.0.bias", "proj.1.weight", "proj.1.bias"))
    sh = e1_shapes.shape_from_state(dataclasses.replace(s, proj_dims=(), d_model=1), pj)
    assert sh.proj_dims == s.proj_dims and sh.d_model == s.d_model
    with pytest.raises(KeyError):
        e1_shapes.canonical_encoder_keys({"something.else": np.zeros(3)})
    with pytest.raises(KeyError):
        e1_shapes.canonical_projector_keys({"norm.weight": np.zeros(3)})


def test_abi_rejects_unsupported_geometry():
    """v1 limits are errors, not silent: head dim != 32, a map that is not whole windows."""
    from markushgrapher_amd.e1 import E1Engine
    from markushgrapher_amd.engine import MgError
    be = get_backend("emu")
    with pytest.raises(MgError, match="outside the supported geometry"):
        E1Engine(dataclasses.replace(PRESETS["tiny"], num_heads=(1, 2, 4)), lib=be.lib, mem=NumpyMem())
    with pytest.raises(MgError, match="outside the supported geometry"):
        E1Engine(dataclasses.replace(PRESETS["tiny"], image_size=80), lib=be.lib, mem=NumpyMem())
    eng = E1Engine(PRESETS["tiny"], lib=be.lib, mem=NumpyMem())
    with pytest.raises(MgError, match="was not loaded"):
        eng.load_state_dict({k: v for k, v in recipe_state_dict(PRESETS["tiny"]).items() if not k.endswith("o_proj.bias")})


# ---- the C ABI against stock and the oracle ---------------------------------------------------------------------------------------
def _check_features(f, g, s, sd, src, B):
    import torch
    from oracle.swin_oracle import SwinOracle
    ref = g["features"] if "features" in g else None
    if ref is not None:
        err = np.abs(f - ref[:B])
        assert err.max() < FEAT_MAX and err.mean() < FEAT_MEAN, (err.max(), err.mean())
    else:
        rows = g["probe_rows"]
        err = np.abs(f[:, rows] - g["features_probe"][:B])
        assert err.max() < FEAT_MAX and err.mean() < FEAT_MEAN, (err.max(), err.mean())
        assert np.abs(f.mean(-1) - g["features_row_mean"][:B]).max() < 5e-3
    with torch.no_grad():
        emu = SwinOracle(s, sd, emulate_bf16=True)
        fe = emu.features(emu.derive_input(src[:B])).numpy()
    assert np.abs(f - fe).max() < EMU_MAX, np.abs(f - fe).max()


@pytest.mark.parametrize("be_name", BACKENDS)
def test_e1_tiny_against_stock_and_oracle(be_name):
    """features vs stock SwinModel (pinned); e1 = projector(features) vs the oracle (the INFERRED part)."""
    g = load_golden("swin_tiny.npz")
    s = PRESETS["tiny"]
    sd = recipe_state_dict(s)
    B = int(g["B"])
    src = synth_pixels(s, B)
    eng = make_e1(be_name, s, sd)
    e1, f = eng.encode(src, want_features=True)
    e1, f = _np(e1), _np(f)
    assert f.shape == (B, s.out_tokens, s.out_dim) and e1.shape == (B, s.out_tokens, s.d_model)
    _check_features(f, g, s, sd, src, B)
    err = np.abs(e1 - g["e1"])
    assert err.max() < 0.02 * float(g["e1_absmax"]) + 0.02, err.max()
    # every row differs from every other (no row was written twice / left out by the window gather and scatter)
    assert len({f[0, t].tobytes() for t in range(s.out_tokens)}) == s.out_tokens


@pytest.mark.parametrize("be_name", BACKENDS)
def test_e1_resize_and_renormalisation(be_name):
    """The INFERRED input derivation: bilinear resize (torch interpolate semantics) + per-channel affine, against the oracle."""
    import torch
    from oracle.swin_oracle import SwinOracle
    s = dataclasses.replace(PRESETS["tiny"], src_image_size=80, **e1_shapes.IMAGENET_RENORM)
    sd = recipe_state_dict(s)
    src = synth_pixels(s, 1)
    eng = make_e1(be_name, s, sd)
    e1, f = eng.encode(src, want_features=True)
    orc = SwinOracle(s, sd, emulate_bf16=True)
    with torch.no_grad():
        fo = orc.features(orc.derive_input(src)).numpy()
    assert np.abs(_np(f) - fo).max() < EMU_MAX


@pytest.mark.parametrize("be_name", BACKENDS)
def test_e1_odd_batch_rows_do_not_meet(be_name):
    """B = 3 (row counts that are not whole 32-row tiles in the last two stages: 48 and 192 rows) against the oracle; the images of a batch
    never meet: the first two rows carry the bits of the B = 2 call."""
    import torch
    from oracle.swin_oracle import SwinOracle
    s = PRESETS["tiny"]
    sd = recipe_state_dict(s)
    src = synth_pixels(s, 3)
    eng = make_e1(be_name, s, sd)
    e3, f3 = (_np(a) for a in eng.encode(src, want_features=True))
    e2, f2 = (_np(a) for a in eng.encode(src[:2], want_features=True))
    assert f3.shape == (3, s.out_tokens, s.out_dim) and np.isfinite(e3).all()
    assert np.array_equal(f3[:2], f2) and np.array_equal(e3[:2], e2)
    orc = SwinOracle(s, sd, emulate_bf16=True)
    with torch.no_grad():
        fo = orc.features(orc.derive_input(src)).numpy()
    assert np.abs(f3 - fo).max() < EMU_MAX


@pytest.mark.gpu
def test_e1_window12_against_stock():
    g = load_golden("swin_w12.npz")
    s = PRESETS["w12"]
    sd = recipe_state_dict(s)
    B = int(g["B"])
    src = synth_pixels(s, B)
    e1, f = make_e1("hip", s, sd).encode(src, want_features=True)
    _check_features(_np(f), g, s, sd, src, B)
    err = np.abs(_np(e1) - g["e1"])
    assert err.max() < 0.02 * float(g["e1_absmax"]) + 0.02, err.max()


@pytest.mark.gpu
def test_e1_swin_b_geometry_against_stock():
    """MolScribe's Swin-B geometry (86.88 M parameters, 384 px -> [B, 144, 1024]) against stock SwinModel's probes; B = 2 here and a
    batch of 32 (the benchmark's) whose first two rows must carry the same bits (rows of a batch never meet)."""
    g = load_golden("swin_b_384.npz")
    s = PRESETS["swin_b_384"]
    sd = recipe_state_dict(s)
    B = int(g["B"])
    src = synth_pixels(s, B)
    eng = make_e1("hip", s, sd)
    e1, f = eng.encode(src, want_features=True)
    e1, f = _np(e1), _np(f)
    assert f.shape == (B, 144, 1024) and e1.shape == (B, 144, 1024)
    _check_features(f, g, s, sd, src, B)
    err = np.abs(e1[:, g["probe_rows"]] - g["e1_probe"])
    assert err.max() < 0.02 * float(g["e1_absmax"]) + 0.02, err.max()
    big = np.concatenate([src] + [synth_pixels(s, 2, seed=7 + i) for i in range(15)])
    e32 = _np(eng.encode(big))
    assert e32.shape == (32, 144, 1024) and np.isfinite(e32).all()
    assert np.array_equal(e32[:2], e1)


# ---- the branch attached to the VTL model (mg_attach_e1): generate() / forward() / the queue forms without `e1=` ---------------------
def _attached(be_name, with_branch=True):
    from tests.backends import make_engine
    from tests.test_oracle_golden import _weights, _inputs
    g = load_golden("g3_trained_tiny.npz")
    shape, sd = _weights(g)
    inp = _inputs(g, shape)
    s1 = PRESETS["tiny"]
    assert s1.d_model == shape.d_model and s1.src_image_size == shape.image_size
    sd1 = recipe_state_dict(s1)
    eng = make_engine(be_name, shape, sd)
    e1e = make_e1(be_name, s1, sd1)
    if with_branch:
        eng.attach_e1(e1e)
    return g, shape, sd, inp, s1, sd1, eng, e1e


@pytest.mark.parametrize("be_name", BACKENDS)
def test_attached_branch_equals_precomputed_tokens_and_oracle(be_name):
    """With the branch attached, a call WITHOUT e1 computes it from pixel_values itself: same bits as passing E1Engine.encode's
    output, greedy ids and teacher-forced logits against the oracle chain SwinOracle.e1 -> Oracle(e1=...) (fusion INFERRED: unpinned)."""
    import torch
    from oracle.swin_oracle import SwinOracle
    from oracle.udop_oracle import Oracle
    g, shape, sd, inp, s1, sd1, eng, e1e = _attached(be_name)
    args = (inp["input_ids"], inp["bbox"], inp["attention_mask"], inp["pixel_values"])
    T = int(g["max_length"])
    e1 = _np(e1e.encode(inp["pixel_values"]))
    ids_own, _, _ = eng.generate(*args, max_length=T)
    ids_pre, _, _ = eng.generate(*args, max_length=T, e1=e1)
    ids_own, ids_pre = eng.mem.numpy(ids_own), eng.mem.numpy(ids_pre)
    assert np.array_equal(ids_own, ids_pre)
    with torch.no_grad():
        e1_ref = SwinOracle(s1, sd1).e1(inp["pixel_values"]).numpy()
    assert np.abs(e1 - e1_ref).max() < 0.02 * np.abs(e1_ref).max() + 0.02
    o = Oracle(shape, sd)
    labels = g["labels"]
    dec_ids = Oracle.shift_right(labels, shape.decoder_start_token_id, shape.pad_token_id).numpy()
    dam = (labels != -100).astype(np.uint8)
    logits, _, _ = eng.forward_logits(*args, dec_ids, dam)
    ref = o.forward(inp["input_ids"], inp["bbox"], inp["pixel_values"], inp["attention_mask"], labels=labels,
                    decoder_attention_mask=dam.astype(np.int64), e1=e1_ref).numpy()
    tol = 0.015 * np.abs(ref).max() + 0.02
    assert np.abs(eng.mem.numpy(logits) - ref).max() < tol
    plain = o.forward(inp["input_ids"], inp["bbox"], inp["pixel_values"], inp["attention_mask"], labels=labels,
                      decoder_attention_mask=dam.astype(np.int64)).numpy()
    assert np.abs(ref - plain).max() > 5 * tol                                   # the tokens matter (the trained ids are robust to them)
    rec = []
    ref_ids = o.greedy(inp["input_ids"], inp["bbox"], inp["pixel_values"], inp["attention_mask"], max_length=T, record=rec, e1=e1_ref)
    compared = 0
    for b in range(ids_own.shape[0]):
        for t in range(1, min(ids_own.shape[1], ref_ids.shape[1])):
            srt = np.sort(rec[t - 1][b].numpy())
            if srt[-1] - srt[-2] < 4 * tol:
                break
            assert ids_own[b, t] == ref_ids[b, t], (b, t)
            compared += 1
    assert compared >= 2 * ids_own.shape[0]
    # beam search and a clone made after the attachment take the same path
    b_own, s_own, _ = eng.generate(*args, num_beams=3, max_length=T)
    b_pre, s_pre, _ = eng.clone().generate(*args, num_beams=3, max_length=T, e1=e1)
    assert np.array_equal(eng.mem.numpy(b_own), eng.mem.numpy(b_pre))
    # detaching gives the plain VTL model back
    eng.attach_e1(None)
    ids0, _, _ = eng.generate(*args, max_length=T)
    assert np.array_equal(eng.mem.numpy(ids0), g["greedy_ids"])


@pytest.mark.parametrize("be_name", BACKENDS)
def test_attached_branch_in_the_queue_forms(be_name):
    """mg_generate_stream / mg_generate_stream_beam with the branch attached: every image's ids equal generate()'s for that image."""
    g, shape, sd, inp, s1, sd1, eng, e1e = _attached(be_name)
    T = int(g["max_length"])
    order = np.array([0, 3, 5, 1, 2, 4, 4, 0])
    q = {k: np.ascontiguousarray(v[order]) for k, v in inp.items()}
    rows, brows = [], []
    for b in range(inp["input_ids"].shape[0]):
        one = {k: v[b:b + 1] for k, v in inp.items()}
        ids, _, _ = eng.generate(one["input_ids"], one["bbox"], one["attention_mask"], one["pixel_values"], max_length=T)
        rows.append(eng.mem.numpy(ids)[0].copy())
        ids, _, _ = eng.generate(one["input_ids"], one["bbox"], one["attention_mask"], one["pixel_values"], num_beams=3, max_length=T)
        brows.append(eng.mem.numpy(ids)[0].copy())
    ids, lens, _ = eng.generate_stream(q["input_ids"], q["bbox"], q["attention_mask"], q["pixel_values"], max_length=T, chunk=3, slots=3, pool_chunks=2)
    ids, lens = eng.mem.numpy(ids), eng.mem.numpy(lens)
    for n, b in enumerate(order):
        row = rows[b]
        e = np.nonzero(row == shape.eos_token_id)[0]
        want = int(e[0]) + 1 if len(e) else len(row)
        assert lens[n] == want and np.array_equal(ids[n, :want], row[:want]), (n, b)
    bids, blens, _, _ = eng.generate_stream_beam(q["input_ids"], q["bbox"], q["attention_mask"], q["pixel_values"], num_beams=3, max_length=T,
                                                 chunk=3, slots=2, pool_chunks=2)
    bids, blens = eng.mem.numpy(bids), eng.mem.numpy(blens)
    for n, b in enumerate(order):
        row = brows[b]
        e = np.nonzero(row == shape.eos_token_id)[0]
        want = int(e[0]) + 1 if len(e) else len(row)
        assert blens[n] == want and np.array_equal(bids[n, :want], row[:want]), (n, b)


def test_attach_rejects_a_branch_of_the_wrong_geometry():
    from markushgrapher_amd.engine import MgError
    g, shape, sd, inp, s1, sd1, eng, e1e = _attached("emu", with_branch=False)
    bad = dataclasses.replace(s1, d_model=128)
    with pytest.raises(MgError, match="mg_attach_e1"):
        eng.attach_e1(make_e1("emu", bad, recipe_state_dict(bad)))
