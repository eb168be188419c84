"""ChemicalOCR stage: oracle vs the stock-Idefics3 golden vectors, and the C ABI `mg_ocr_*` against both.

Fixtures (tools/make_golden_ocr.py, minted from stock transformers Idefics3ForConditionalGeneration on recipe weights):
  ocr_tiny.npz        every code path at a size the emulator finishes in seconds; B = 3, rows end at different steps (EOS)
  ocr_smoldocling.npz SmolDocling-256M geometry (INFERRED for the reference's checkpoint), B = 2, 8 greedy steps

Tolerances: the HIP path keeps weights and GEMM / attention operands in bf16 with fp32 accumulation and an fp32 residual
stream; the golden vectors are fp32.
  * image features: max-abs error < 5 % of their mean magnitude (FEAT_TOL below)
  * logits: max-abs error < 1.5 % of the fixture's max |logit| + 0.02 (the main path's rule)
  * ids: equal wherever stock's top-1 / top-2 margin exceeds 4x the logit tolerance; after the first low-margin step of a
    row the comparison of that row stops (its continuation legitimately differs)."""
import numpy as np
import pytest

from markushgrapher_amd.ocr_shapes import PRESETS, recipe_state_dict, synth_inputs
from tests.backends import get_backend, NumpyMem
from tests.conftest import load_golden

BACKENDS = [pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)]


def logit_tol(absmax):
    return 0.015 * float(absmax) + 0.02


def _setup(name):
    import dataclasses
    g = load_golden(f"ocr_{name}.npz")
    s = dataclasses.replace(PRESETS[str(g["shape"])], eos_token_id=int(g["eos_token_id"]))
    sd = recipe_state_dict(s, gain=float(g["gain"]))
    ids, pix = synth_inputs(s, int(g["B"]), n_img=int(g["n_img"]) if "n_img" in g else 1)
    assert np.array_equal(ids, g["input_ids"])
    return g, s, sd, ids, pix


def make_ocr(be_name, s, sd):
    from markushgrapher_amd.ocr import OcrEngine
    be = get_backend(be_name)
    eng = OcrEngine(s, lib=be.lib, mem=NumpyMem()) if be_name == "emu" else OcrEngine(s)
    return eng.load_state_dict(sd)


def test_oracle_reproduces_stock_tiny():
    """The CPU restatement against the stock outputs stored in the fixture (full logits, image features, ids with EOS)."""
    import torch
    from oracle.ocr_oracle import OcrOracle
    g, s, sd, ids, pix = _setup("tiny")
    orc = OcrOracle(s, sd)
    with torch.no_grad():
        feats = orc.image_features(pix).numpy()
        logits = orc.forward(ids, pix).numpy()
        new = orc.generate(ids, pix, int(g["new_tokens"])).numpy()
    assert np.abs(feats - g["feats"]).max() < 2e-4
    assert np.abs(logits - g["logits"]).max() < 2e-3
    assert np.array_equal(new, g["new_ids"])
    # rows end at different steps, finished rows are padded
    assert (g["new_ids"][0] == s.pad_token_id).any() and not (g["new_ids"] == s.pad_token_id).all(axis=1).any()


def test_recipe_and_spec_are_consistent():
    from markushgrapher_amd.ocr_shapes import state_dict_spec
    for name in ("tiny", "smoldocling"):
        s = PRESETS[name]
        spec = state_dict_spec(s)
        assert len({k for k, _, _ in spec}) == len(spec)
        n = sum(int(np.prod(shape)) for _, shape, _ in spec)
        assert n > 0
    assert 2.4e8 < sum(int(np.prod(shape)) for _, shape, _ in state_dict_spec(PRESETS["smoldocling"])) < 2.9e8   # "256M"


def _check_generate(g, s, new, cap):
    tol = logit_tol(g["logits_absmax"])
    ref_ids, margin = g["new_ids"], g["step_margin"]
    vals, idx = g["step_top8_val"], g["step_top8_idx"]
    n = ref_ids.shape[1]
    assert new.shape[1] <= int(g["new_tokens"])
    checked = 0
    for b in range(ref_ids.shape[0]):
        alive = True
        for t in range(n):
            if not alive or ref_ids[b, t] == s.pad_token_id and t > 0 and (ref_ids[b, :t] == s.eos_token_id).any():
                break
            if t < cap.shape[0]:
                got = cap[t, b, idx[b, t]]
                assert np.abs(got - vals[b, t]).max() < tol, (b, t, got, vals[b, t])
            if margin[b, t] > 4 * tol:
                assert t < new.shape[1] and new[b, t] == ref_ids[b, t], (b, t, new[b].tolist(), ref_ids[b].tolist())
                checked += 1
            else:
                alive = t < new.shape[1] and new[b, t] == ref_ids[b, t]
    assert checked >= ref_ids.shape[0]


@pytest.mark.parametrize("be_name", BACKENDS)
def test_ocr_tiny_matches_stock(be_name):
    g, s, sd, ids, pix = _setup("tiny")
    eng = make_ocr(be_name, s, sd)
    feats = eng.mem.numpy(eng.image_features(pix[:, 0]))
    FEAT_TOL = 0.05 * float(g["feats_abs_mean"])
    assert np.abs(feats - g["feats"]).max() < FEAT_TOL, np.abs(feats - g["feats"]).max()
    logits = eng.mem.numpy(eng.forward_logits(ids, pix))
    tol = logit_tol(g["logits_absmax"])
    assert np.abs(logits - g["logits"]).max() < tol, (np.abs(logits - g["logits"]).max(), tol)
    new, cap = eng.generate(ids, pix, int(g["new_tokens"]), capture_steps=int(g["new_tokens"]))
    new, cap = eng.mem.numpy(new), eng.mem.numpy(cap)
    _check_generate(g, s, new, cap)
    # the run stops when every row has ended: same number of columns as stock, finished rows padded
    assert new.shape == g["new_ids"].shape
    fin = g["new_ids"] == s.pad_token_id
    assert np.array_equal(new[fin], g["new_ids"][fin])


@pytest.mark.parametrize("be_name", BACKENDS)
def test_ocr_rejects_bad_inputs(be_name):
    from markushgrapher_amd.engine import MgError
    g, s, sd, ids, pix = _setup("tiny")
    eng = make_ocr(be_name, s, sd)
    bad = ids.copy()
    bad[0, 0] = s.vocab + 5
    with pytest.raises(MgError):
        eng.forward_logits(bad, pix)
    fewer = ids.copy()
    fewer[1, np.argmax(ids[1] == s.image_token_id)] = 3          # one <image> token short
    with pytest.raises(MgError):
        eng.generate(fewer, pix, 2)


@pytest.mark.gpu
def test_ocr_smoldocling_shape_matches_stock():
    g, s, sd, ids, pix = _setup("smoldocling")
    eng = make_ocr("hip", s, sd)
    feats = eng.mem.numpy(eng.image_features(pix[:, 0]))
    step = max(1, feats.shape[1] // 4)
    FEAT_TOL = 0.05 * float(g["feats_abs_mean"])
    assert np.abs(feats[:, ::step] - g["feats_probe"]).max() < FEAT_TOL, np.abs(feats[:, ::step] - g["feats_probe"]).max()
    assert np.abs(feats.astype(np.float64).sum(axis=(1, 2)) - g["feats_checksum"]).max() < FEAT_TOL * feats.shape[1] * feats.shape[2] * 0.02
    logits = eng.mem.numpy(eng.forward_logits(ids, pix))
    tol = logit_tol(g["logits_absmax"])
    got = np.take_along_axis(logits, g["logits_top8_idx"], axis=-1)
    assert np.abs(got - g["logits_top8_val"]).max() < tol, (np.abs(got - g["logits_top8_val"]).max(), tol)
    new, cap = eng.generate(ids, pix, int(g["new_tokens"]), capture_steps=int(g["new_tokens"]))
    _check_generate(g, s, eng.mem.numpy(new), eng.mem.numpy(cap))


@pytest.mark.gpu
def test_ocr_graph_replay_equals_eager():
    """The captured decode step (positions read from the device step counter) produces the ids of the eager launches, bit for bit,
    also when it is replayed by a second call and when rows end early."""
    g, s, sd, ids, pix = _setup("tiny")
    eng = make_ocr("hip", s, sd)
    n = int(g["new_tokens"])
    eager, _ = eng.generate(ids, pix, n, capture_steps=1)          # instrumented calls launch eagerly
    eager = eng.mem.numpy(eager).copy()
    for _ in range(2):
        got, _ = eng.generate(ids, pix, n)
        assert np.array_equal(eng.mem.numpy(got), eager)
    g2, s2, sd2, ids2, pix2 = _setup("smoldocling")
    eng2 = make_ocr("hip", s2, sd2)
    a, _ = eng2.generate(ids2, pix2, 24, capture_steps=1)
    b, _ = eng2.generate(ids2, pix2, 24)
    assert np.array_equal(eng2.mem.numpy(a), eng2.mem.numpy(b))


def _stock_tiny(tmp_path, s, sd, bf16=False):
    import torch
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from make_golden_ocr import stock_model
    m = stock_model(s, sd)
    if bf16:
        m = m.to(torch.bfloat16)          # the recipe weights are bf16-exact: the checkpoint holds the same values in half the bytes
    m.save_pretrained(str(tmp_path), safe_serialization=True)
    return m


def test_shape_from_hf_config_roundtrip(tmp_path):
    """config.json written by the stock class -> OcrShape: what from_pretrained builds the engine from."""
    import dataclasses
    from markushgrapher_amd.ocr import shape_from_hf_config
    g, s, sd, ids, pix = _setup("tiny")
    _stock_tiny(tmp_path, s, sd)
    got = shape_from_hf_config(str(tmp_path))
    assert dataclasses.asdict(got) == dataclasses.asdict(s)


@pytest.mark.gpu
@pytest.mark.parametrize("bf16", [False, True])
def test_ocr_model_from_pretrained_generates_like_stock(tmp_path, bf16):
    """The reference-facing surface: OcrModel.from_pretrained(dir).generate(**inputs, max_new_tokens=, do_sample=False) returns
    [prompt | new tokens] as the stock model does (chemical_ocr.py:375-386)."""
    import torch
    from markushgrapher_amd.ocr import OcrModel
    g, s, sd, ids, pix = _setup("tiny")
    _stock_tiny(tmp_path, s, sd, bf16=bf16)
    model = OcrModel.from_pretrained(str(tmp_path)).eval()
    tid, tpix = torch.from_numpy(ids), torch.from_numpy(pix)
    out = model.generate(input_ids=tid, attention_mask=torch.ones_like(tid), pixel_values=tpix,
                         pixel_attention_mask=torch.ones(ids.shape[0], 1, s.image_size, s.image_size, dtype=torch.bool),
                         max_new_tokens=int(g["new_tokens"]), do_sample=False).cpu().numpy()
    assert np.array_equal(out[:, :ids.shape[1]], ids)
    new = out[:, ids.shape[1]:]
    ref, margin = g["new_ids"], g["step_margin"]
    tol = logit_tol(g["logits_absmax"])
    for b in range(ref.shape[0]):
        for t in range(min(new.shape[1], ref.shape[1])):
            if margin[b, t] <= 4 * tol:
                break
            assert new[b, t] == ref[b, t]


@pytest.mark.parametrize("be_name", BACKENDS)
def test_ocr_two_frames_per_page(be_name):
    """pixel_values [B][2][3][I][I]: a page the processor split into two frames - 2 x image_seq_len <image> tokens per sequence,
    features scattered in frame order (inputs_merger's masked_scatter)."""
    g, s, sd, ids, pix = _setup("tiny2")
    assert pix.shape[1] == 2 and (ids == s.image_token_id).sum(axis=1).tolist() == [2 * s.image_seq_len] * ids.shape[0]
    eng = make_ocr(be_name, s, sd)
    logits = eng.mem.numpy(eng.forward_logits(ids, pix))
    tol = logit_tol(g["logits_absmax"])
    assert np.abs(logits - g["logits"]).max() < tol
    n = int(g["new_tokens"])
    new, cap = eng.generate(ids, pix, n, capture_steps=n)
    _check_generate(g, s, eng.mem.numpy(new), eng.mem.numpy(cap))


@pytest.mark.gpu
def test_ocr_more_than_one_row_tile():
    """35 sequences = two 32-row tiles: the decode step then runs its projections one row tile per workgroup and without the
    [down_proj | next QKV] pair launch.  Checked against the oracle (pinned on stock by the fixtures above), margin rule."""
    import torch
    from oracle.ocr_oracle import OcrOracle
    g, s, sd, _, _ = _setup("tiny")
    B, n = 35, 6
    ids, pix = synth_inputs(s, B)
    eng = make_ocr("hip", s, sd)
    new, cap = eng.generate(ids, pix, n, capture_steps=n)
    new, cap = eng.mem.numpy(new), eng.mem.numpy(cap)
    with torch.no_grad():
        ref, sc = OcrOracle(s, sd).generate(ids, pix, n, return_logits=True)
    ref, sc = ref.numpy(), sc.numpy()
    tol = logit_tol(np.abs(sc).max())
    srt = np.sort(sc, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    checked = 0
    for b in range(B):
        for t in range(min(new.shape[1], ref.shape[1])):
            if t > 0 and (ref[b, :t] == s.eos_token_id).any():
                break
            assert np.abs(cap[t, b] - sc[b, t]).max() < tol, (b, t)
            if margin[b, t] <= 4 * tol:
                break
            assert new[b, t] == ref[b, t], (b, t)
            checked += 1
    assert checked >= B // 2          # (every row's logits were compared at least at step 0; ids wherever the margin allows)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_ocr_padded_frames(be_name):
    """Non-square pages: the processor pads the frame and masks the padding.  Masked patches take position id 0 and are not attended
    as keys in the vision tower (modeling_idefics3.py:128-172, 462-505); the patch grid comes from OcrEngine.patch_inputs."""
    from markushgrapher_amd.ocr_shapes import synth_pixel_mask
    g, s, sd, ids, pix = _setup("tiny3")
    pam = synth_pixel_mask(s, int(g["B"]), 1)
    assert not pam.all() and pam[0, 0].all()
    pix = np.where(pam[:, :, None], pix, np.float32(0.0)).astype(np.float32)
    eng = make_ocr(be_name, s, sd)
    feats = eng.mem.numpy(eng.image_features(pix[:, 0], pam[:, 0]))
    step = max(1, feats.shape[1] // 4)
    assert np.abs(feats[:, ::step] - g["feats_probe"]).max() < 0.05 * float(g["feats_abs_mean"])
    logits = eng.mem.numpy(eng.forward_logits(ids, pix, pam))
    tol = logit_tol(g["logits_absmax"])
    got = np.take_along_axis(logits, g["logits_top8_idx"], axis=-1)
    assert np.abs(got - g["logits_top8_val"]).max() < tol
    n = int(g["new_tokens"])
    new, cap = eng.generate(ids, pix, n, capture_steps=n, pixel_attention_mask=pam)
    _check_generate(g, s, eng.mem.numpy(new), eng.mem.numpy(cap))
    # and the mask matters: without it the features of the padded frames differ
    plain = eng.mem.numpy(eng.image_features(pix[:, 0]))
    assert np.abs(plain[1:] - feats[1:]).max() > 10 * 0.05 * float(g["feats_abs_mean"]) or np.abs(plain[1:] - feats[1:]).max() > 0.01


def test_patch_inputs_rejects_padding_frames():
    """A fully masked frame (the processor's padding image for sequences with fewer frames) has no patch grid: loud error, not NaNs."""
    from markushgrapher_amd.engine import MgError
    from markushgrapher_amd.ocr import OcrEngine
    s = PRESETS["tiny"]

    class _Mem:                                    # patch_inputs only converts its two results through the memory provider
        def asarray(self, x, dtype):
            return np.asarray(x, dtype)
    eng = OcrEngine.__new__(OcrEngine)
    eng.shape, eng.mem = s, _Mem()
    m = np.ones((2, 1, s.image_size, s.image_size), bool)
    assert eng.patch_inputs(m) == (None, None)
    m[1] = False
    with pytest.raises(MgError):
        eng.patch_inputs(m)
    m[1, 0, :20, :40] = True
    pos, msk = eng.patch_inputs(m)
    assert msk.shape == (2, s.patches) and msk[0].all() and msk[1].sum() == 2 * 3 and pos[1][msk[1] == 0].max() == 0


@pytest.mark.parametrize("be_name", BACKENDS)
def test_ocr_several_stop_tokens(be_name):
    """generation_config.json may list several EOS ids: a row ends on ANY of them.  The second stop id is a token the
    golden run emits mid-sequence with a safe margin, so that row must now end there (pad afterwards, fewer columns overall)."""
    import dataclasses
    import torch
    from oracle.ocr_oracle import OcrOracle
    g, s, sd, ids, pix = _setup("tiny")
    n = int(g["new_tokens"])
    tonp = lambda x: np.asarray(x if isinstance(x, np.ndarray) else x.cpu().numpy())
    base = tonp(make_ocr(be_name, s, sd).generate(ids, pix, n)[0])
    b0, t0 = 1, 3                                               # a row that does not end by itself; its 4th new token becomes a stop id
    extra = int(base[b0, t0])
    assert extra not in (s.eos_token_id, s.pad_token_id) and not (base[b0, :t0] == extra).any()
    s2 = dataclasses.replace(s, eos_extra=(extra,))
    new = tonp(make_ocr(be_name, s2, sd).generate(ids, pix, n)[0])
    for b in range(base.shape[0]):                              # every row: unchanged up to its first stop token, pad afterwards
        hits = np.nonzero((base[b] == extra) | (base[b] == s.eos_token_id))[0]
        end = int(hits[0]) if len(hits) else base.shape[1] - 1
        end = min(end, new.shape[1] - 1)
        assert np.array_equal(new[b, :end + 1], base[b, :end + 1]) and (new[b, end + 1:] == s.pad_token_id).all(), (b, new[b], base[b])
    assert new[b0, t0] == extra
    with torch.no_grad():                                       # the oracle states the same rule
        o0 = OcrOracle(s, sd).generate(ids, pix, n).numpy()
        e2 = int(o0[b0, t0])
        o2 = OcrOracle(dataclasses.replace(s, eos_extra=(e2,)), sd).generate(ids, pix, n).numpy()
    assert o2[b0, t0] == e2 and (o2[b0, t0 + 1:] == s.pad_token_id).all() and np.array_equal(o2[b0, :t0], o0[b0, :t0])
    from markushgrapher_amd.engine import MgError
    with pytest.raises(MgError, match="stop tokens"):
        make_ocr(be_name, dataclasses.replace(s, eos_extra=(5, 6, 7, 8)), sd)


def test_ocr_shape_from_hf_config_reads_generation_config(tmp_path):
    """Stop tokens come from generation_config.json when the checkpoint has one (what model.generate() uses), else config.json."""
    import json
    from markushgrapher_amd.ocr import shape_from_hf_config
    from markushgrapher_amd.engine import MgError
    cfg = {"vision_config": {"hidden_size": 768, "intermediate_size": 3072, "num_hidden_layers": 12, "num_attention_heads": 12,
                             "image_size": 512, "patch_size": 16},
           "text_config": {"hidden_size": 576, "intermediate_size": 1536, "num_hidden_layers": 30, "num_attention_heads": 9,
                           "num_key_value_heads": 3, "vocab_size": 49280, "rope_theta": 100000.0, "rms_norm_eps": 1e-5},
           "scale_factor": 4, "image_token_id": 49190, "eos_token_id": 49279, "pad_token_id": 2}
    d = tmp_path / "ckpt"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(cfg))
    s = shape_from_hf_config(str(d))
    assert s.eos_token_id == 49279 and s.eos_extra == () and s.t_kv_heads == 3 and s.image_seq_len == 64
    (d / "generation_config.json").write_text(json.dumps({"eos_token_id": [49279, 49154], "max_new_tokens": 40}))
    s = shape_from_hf_config(str(d))
    assert s.eos_token_id == 49279 and s.eos_extra == (49154,)
    (d / "generation_config.json").write_text(json.dumps({"eos_token_id": 49154}))
    assert shape_from_hf_config(str(d)).eos_token_id == 49154
    (d / "generation_config.json").write_text(json.dumps({"eos_token_id": [1, 2, 3, 4, 5]}))
    with pytest.raises(MgError, match="stop tokens"):
        shape_from_hf_config(str(d))


@pytest.mark.gpu
@pytest.mark.parametrize("B", [35, 128])
def test_ocr_bench_batches_at_smoldocling_geometry(B):
    """The OCR bench shapes under an oracle check: SmolDocling-256M geometry at 35 pages (two row tiles) and
    128 pages (four: `gemm_rows_split_kernel` / `gemm_rows_resid_rowsplit_kernel` at production dimensions), 16 greedy steps.
    Rows are independent, so the oracle runs on a subset that spans every row tile (first / last row of each tile, plus the odd
    tail rows at 35); per-step logits within the logit tolerance, ids under the margin rule."""
    import dataclasses
    import torch
    from oracle.ocr_oracle import OcrOracle
    s = dataclasses.replace(PRESETS["smoldocling"], eos_token_id=-1)          # EOS cannot occur: every row runs all steps
    g = load_golden("ocr_smoldocling.npz")
    sd = recipe_state_dict(s, gain=float(g["gain"]))
    n = 16
    ids, pix = synth_inputs(s, B)
    eng = make_ocr("hip", s, sd)
    new, cap = eng.generate(ids, pix, n, capture_steps=n)
    new, cap = eng.mem.numpy(new), eng.mem.numpy(cap)              # [B, n], [n, B, V]
    assert new.shape == (B, n)
    rows = sorted({r for t in range((B + 31) // 32) for r in (32 * t, min(32 * t + 31, B - 1))} | {B - 2, B - 1})
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        ref, sc = OcrOracle(s, sd).generate(ids[rows], pix[rows], n, return_logits=True)
    ref, sc = ref.numpy(), sc.numpy()                              # [R, n], [R, n, V]
    tol = logit_tol(np.abs(sc).max())
    srt = np.sort(sc, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    steps = ids_eq = 0
    for i, b in enumerate(rows):
        for t in range(n):
            # same prefix so far: this step's logits are comparable
            assert np.abs(cap[t, b] - sc[i, t]).max() < tol, (b, t, float(np.abs(cap[t, b] - sc[i, t]).max()), tol)
            steps += 1
            if new[b, t] != ref[i, t]:
                assert margin[i, t] <= 4 * tol, (b, t, float(margin[i, t]))      # only a near-tie may part the two runs
                break                                              # the continuation of this row legitimately differs from here
            ids_eq += 1
    assert steps >= 3 * len(rows) and ids_eq >= 2 * len(rows), (steps, ids_eq)


def test_shape_from_hf_config_smoldocling_written_by_stock(tmp_path):
    """A SmolDocling-256M-style config.json written by the stock Idefics3Config (what a ChemicalOCR checkpoint directory holds, geometry
    INFERRED) -> shape_from_hf_config gives the preset every OCR bench / test of this repository runs on."""
    import dataclasses
    from transformers import Idefics3Config
    from markushgrapher_amd.ocr import shape_from_hf_config
    s = PRESETS["smoldocling"]
    cfg = Idefics3Config(
        vision_config=dict(hidden_size=s.v_hidden, intermediate_size=s.v_inter, num_hidden_layers=s.v_layers, num_attention_heads=s.v_heads,
                           image_size=s.image_size, patch_size=s.patch_size, num_channels=3, hidden_act="gelu_pytorch_tanh", layer_norm_eps=s.v_eps),
        text_config=dict(model_type="llama", hidden_size=s.t_hidden, intermediate_size=s.t_inter, num_hidden_layers=s.t_layers,
                         num_attention_heads=s.t_heads, num_key_value_heads=s.t_kv_heads, vocab_size=s.vocab, rms_norm_eps=s.rms_eps,
                         max_position_embeddings=8192, rope_theta=s.rope_theta, tie_word_embeddings=s.tie_word_embeddings,
                         pad_token_id=s.pad_token_id, bos_token_id=0, eos_token_id=s.eos_token_id, head_dim=64),
        scale_factor=s.scale_factor, image_token_id=s.image_token_id, pad_token_id=s.pad_token_id, tie_word_embeddings=s.tie_word_embeddings)
    cfg.save_pretrained(str(tmp_path))
    got = shape_from_hf_config(str(tmp_path))
    assert dataclasses.asdict(got) == dataclasses.asdict(s)
    assert got.image_seq_len == 64 and got.patches == 1024


@pytest.mark.gpu
def test_ocr_max_new_tokens_4096_capacity():
    """The reference's setting, generate(max_new_tokens=4096) (ref: ocr/chemical_ocr.py:381-385): 4096 new tokens behind a 16-token prompt -
    KV caches of 4112 positions, the rotation table beyond 4096, 4095 graph replays.  Random weights cannot be compared that far (the
    first near-tie parts two runs), so the tiny model's lm_head is SCRIPTED to walk a fixed cycle of ~300 tokens (ocr_shapes.
    scripted_state_dict, margins ~100 x the bf16 noise): every one of the 3 x 4096 ids is known in advance.  The oracle confirms the
    script on the first 48 steps."""
    import dataclasses
    import torch
    from markushgrapher_amd import synth
    from markushgrapher_amd.ocr_shapes import scripted_prompts, scripted_state_dict
    from oracle.ocr_oracle import OcrOracle
    s = dataclasses.replace(PRESETS["tiny"], eos_token_id=1)
    cyc = [i for i in range(3, s.vocab) if i not in (s.eos_token_id, s.pad_token_id, s.image_token_id)][:296]
    starts = [cyc[-1], cyc[99], cyc[199]]                       # three rows enter the cycle at different points
    sd = scripted_state_dict(s, [cyc + [s.eos_token_id]], [cyc[-1]])
    head = sd["lm_head.weight"]
    head[s.eos_token_id] = synth.round_bf16(synth.uniform_pm1("cap/eos", head[0].shape, 1) * np.float32(0.05))    # the cycle never ends
    ids = scripted_prompts(s, [cyc], starts, 10)
    _, pix = synth_inputs(s, 3)
    n = 4096
    want = np.array([[cyc[(cyc.index(st) + 1 + t) % len(cyc)] for t in range(n)] for st in starts])
    with torch.no_grad():
        ref = OcrOracle(s, sd).generate(ids, pix, 48).numpy()
    assert np.array_equal(ref, want[:, :48])
    eng = make_ocr("hip", s, sd)
    new, _ = eng.generate(ids, pix, n)
    new = eng.mem.numpy(new)
    assert new.shape == (3, n)
    bad = np.nonzero(new != want)
    assert len(bad[0]) == 0, (bad[0][:5], bad[1][:5])


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("slots,chunk", [(2, 3), (3, 8), (5, 2)])
def test_ocr_queue_form_equals_batch_form(be_name, slots, chunk):
    """mg_ocr_generate_stream: 8 pages (the fixture's 3, repeated in a different order) through 2 / 3 / 5 decode rows: every page's new
    ids and length equal what generate() returns for that page (rows end at different steps in the fixture: one emits EOS early)."""
    g, s, sd, ids, pix = _setup("tiny")
    eng = make_ocr(be_name, s, sd)
    n = int(g["new_tokens"])
    base, _ = eng.generate(ids, pix, n)
    base = np.asarray(eng.mem.numpy(base)).copy()                 # [3, cols]
    order = np.array([2, 0, 1, 1, 2, 0, 0, 2])
    new, lens, steps = eng.generate_stream(ids[order], pix[order], n, slots=slots, chunk=chunk)
    new, lens = np.asarray(eng.mem.numpy(new)).copy(), np.asarray(eng.mem.numpy(lens)).copy()
    total = 0
    for k, b in enumerate(order):
        row = base[b]
        e = np.nonzero(row == s.eos_token_id)[0]
        want = int(e[0]) + 1 if len(e) else n
        assert lens[k] == want, (k, b, lens[k], want)
        assert np.array_equal(new[k, :want], row[:want]) and np.all(new[k, want:] == s.pad_token_id)
        total += want - 1
    assert steps <= -(-total // slots) + len(order) + 16
    # the same engine (same workspace, same captured step graph key but for `chunk`) with another chunk: the decode rows, K/V pages
    # and slot table are carved behind the prefill region `chunk` sizes, so the captured graph must not be replayed
    new_b, lens_b, _ = eng.generate_stream(ids[order], pix[order], n, slots=slots, chunk=chunk + 3)
    assert np.array_equal(np.asarray(eng.mem.numpy(new_b)), new) and np.array_equal(np.asarray(eng.mem.numpy(lens_b)), lens)
    new_c, lens_c, _ = eng.generate_stream(ids[order], pix[order], n, slots=slots, chunk=chunk)
    assert np.array_equal(np.asarray(eng.mem.numpy(new_c)), new) and np.array_equal(np.asarray(eng.mem.numpy(lens_c)), lens)
    # a page whose FIRST token is a stop token never takes a slot: make the first token of page 0 a second stop id
    import dataclasses
    s2 = dataclasses.replace(s, eos_extra=(int(base[0, 0]),))
    eng2 = make_ocr(be_name, s2, sd)
    new2, lens2, _ = eng2.generate_stream(ids[order], pix[order], n, slots=slots, chunk=chunk)
    new2, lens2 = np.asarray(eng2.mem.numpy(new2)), np.asarray(eng2.mem.numpy(lens2))
    for k, b in enumerate(order):
        if b == 0:
            assert lens2[k] == 1 and new2[k, 0] == base[0, 0] and np.all(new2[k, 1:] == s.pad_token_id)


def test_oracle_left_padded_batch_reproduces_stock():
    """ocr_tiny_ragged.npz: stock Idefics3 generate() on a LEFT-PADDED batch of three prompts of different lengths; the oracle restates it
    as every row alone without its padding (generate_padded)."""
    import torch
    from oracle.ocr_oracle import OcrOracle
    g = load_golden("ocr_tiny_ragged.npz")
    s = PRESETS["tiny"]
    sd = recipe_state_dict(s, gain=float(g["gain"]))
    _, pix = synth_inputs(s, int(g["B"]))
    with torch.no_grad():
        new, lg = OcrOracle(s, sd).generate_padded(g["input_ids"], g["attention_mask"], pix, int(g["new_tokens"]), return_logits=True)
    assert np.array_equal(new.numpy(), g["new_ids"])
    top = np.take_along_axis(lg.numpy(), g["step_top8_idx"], axis=-1)
    assert np.abs(top - g["step_top8_val"]).max() < 2e-4 * max(1.0, float(np.abs(g["step_top8_val"]).max()))


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("slots,chunk", [(2, 3), (4, 2)])
def test_ocr_prompts_of_different_lengths(be_name, slots, chunk):
    """mg_ocr_generate_stream_ragged: the left-padded batch of the stock fixture (prompt lengths L, L - 3, L - 5), repeated in another order, through the
    queue form: every page's new ids equal stock's (all margins of the fixture > 0.07), and equal what the equal-length path returns for that page's
    prompt alone."""
    g = load_golden("ocr_tiny_ragged.npz")
    s = PRESETS["tiny"]
    sd = recipe_state_dict(s, gain=float(g["gain"]))
    B, n = int(g["B"]), int(g["new_tokens"])
    _, pix = synth_inputs(s, B)
    eng = make_ocr(be_name, s, sd)
    order = np.array([1, 2, 0, 2, 1])
    new, lens, _ = eng.generate_stream(g["input_ids"][order], pix[order], n, slots=slots, chunk=chunk, attention_mask=g["attention_mask"][order])
    new, lens = np.asarray(eng.mem.numpy(new)).copy(), np.asarray(eng.mem.numpy(lens)).copy()
    assert np.all(lens == n) and np.array_equal(new, g["new_ids"][order])
    for b in range(B):                                   # the same prompt alone, unpadded, through the equal-length batch form
        p = int((g["attention_mask"][b] == 0).sum())
        alone, _ = eng.generate(g["input_ids"][b:b + 1, p:], pix[b:b + 1], n)
        assert np.array_equal(np.asarray(eng.mem.numpy(alone)), g["new_ids"][b:b + 1])
    # not a left-padded row: refused on the host; a length outside [1, L]: refused by the library
    bad = g["attention_mask"].copy(); bad[1, -1] = 0
    with pytest.raises(ValueError, match="left-padded"):
        eng.generate_stream(g["input_ids"], pix, n, slots=2, chunk=2, attention_mask=bad)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_ocr_clone_is_a_second_context_on_the_same_weights(be_name):
    """mg_ocr_clone: batch and queue form through a clone equal the source's (own workspace and captured graphs); on the GPU the two
    contexts also run at the same time from two host threads on two streams."""
    g, s, sd, ids, pix = _setup("tiny")
    eng = make_ocr(be_name, s, sd)
    ctx = eng.clone()
    n = int(g["new_tokens"])
    base = np.asarray(eng.mem.numpy(eng.generate(ids, pix, n)[0])).copy()
    got = np.asarray(ctx.mem.numpy(ctx.generate(ids, pix, n)[0])).copy()
    assert np.array_equal(base, got)
    order = np.array([2, 0, 1, 1, 0])
    q0 = eng.generate_stream(ids[order], pix[order], n, slots=2, chunk=3)
    q1 = ctx.generate_stream(ids[order], pix[order], n, slots=2, chunk=3)
    assert np.array_equal(eng.mem.numpy(q0[0]), ctx.mem.numpy(q1[0])) and np.array_equal(eng.mem.numpy(q0[1]), ctx.mem.numpy(q1[1]))
    if be_name == "hip":
        import threading
        import torch
        from markushgrapher_amd.inflight import shared_streams
        sts = shared_streams(torch, eng.mem.device, 2)
        ids_d, pix_d = torch.from_numpy(np.ascontiguousarray(ids)).cuda(), torch.from_numpy(np.ascontiguousarray(pix)).cuda()
        torch.cuda.synchronize()
        outs = [None, None]

        def work(i, e):
            with torch.cuda.device(sts[i].device), torch.cuda.stream(sts[i]):
                for _ in range(6):
                    o = e.generate(ids_d, pix_d, n)[0]
                sts[i].synchronize()
                outs[i] = o.cpu().numpy()
        th = [threading.Thread(target=work, args=(i, e)) for i, e in enumerate((eng, ctx))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert np.array_equal(outs[0], base) and np.array_equal(outs[1], base)
    ctx.close()
    assert np.array_equal(np.asarray(eng.mem.numpy(eng.generate(ids, pix, n)[0])), base)
