"""The HuggingFace-style boundary (markushgrapher_amd/modeling.py): names and behaviours the reference relies on
(ref: markushgrapher/core/common/begin.py:105-172, utils/ocsr/utils_evaluation.py:151-175,269-285,
core/trainers/curriculumTrainer.py:648-656, utils/model/utils_model_loading.py:6-46)."""
import os

import numpy as np
import pytest
import torch

from markushgrapher_amd import synth
from markushgrapher_amd.modeling import MarkushgrapherConfig, MarkushgrapherForConditionalGeneration
from tests.conftest import load_golden, GOLDEN


def tiny_model():
    shape = synth.SHAPES["tiny"]
    cfg = MarkushgrapherConfig(**shape.to_dict())
    m = MarkushgrapherForConditionalGeneration(cfg)
    sd = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "g3_weights.npz")).items()}
    for a, c in synth.tied_aliases(shape).items():
        sd[a] = sd[c]
    missing, unexpected = m.load_state_dict(sd)
    assert not missing and not unexpected, (missing, unexpected)
    return m.eval(), shape


def test_surface_names_and_state_dict(tmp_path):
    m, shape = tiny_model()
    # attributes the reference touches
    for attr in ("encoder", "decoder", "lm_head", "generate", "forward", "get_encoder", "init_molscribe_weights", "safe_load",
                 "device", "config"):
        assert hasattr(m, attr)
    assert hasattr(m.encoder, "molscribe_encoder") and hasattr(m.encoder, "molscribe_projector")
    m.config.image_size = shape.image_size
    m.config.architecture_variant = "me-lf-stack-1"
    m.config.output_attentions = True
    keys = set(m.state_dict().keys())
    for k, _, _ in synth.state_dict_spec(shape):
        assert k in keys, k
    assert "lm_head.weight" in keys
    # sub-module state dicts as saved by the reference (utils_model_loading.py:23-41)
    assert any(k.startswith("block.0.layer.1.EncDecAttention") for k in m.decoder.state_dict())
    assert list(m.lm_head.state_dict().keys()) == ["weight"]
    assert len(list(m.parameters())) > 10
    # save / from_pretrained round trip
    m.save_pretrained(str(tmp_path))
    cfg = MarkushgrapherConfig.from_pretrained(str(tmp_path))
    cfg.image_size = shape.image_size
    m2 = MarkushgrapherForConditionalGeneration.from_pretrained(str(tmp_path), config=cfg)
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k


def test_cpu_device_fails_loudly():
    m, shape = tiny_model()
    g = load_golden("g3_trained_tiny.npz")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.generate(input_ids=torch.from_numpy(g["input_ids"]), bbox=torch.from_numpy(g["bbox"]),
                   pixel_values=torch.from_numpy(g["pixel_values"]), num_beams=1, max_length=8)


@pytest.mark.gpu
def test_generate_and_forward_like_the_reference_calls_them():
    m, shape = tiny_model()
    m = m.to("cuda")
    g = load_golden("g3_trained_tiny.npz")
    dev = m.device
    # ref: utils_evaluation.py:151-175 — per-sample tensors, attention_mask deleted, labels left in the kwargs
    for b in range(g["input_ids"].shape[0]):
        n = int(g["attention_mask"][b].sum())
        enc = {"input_ids": torch.from_numpy(g["input_ids"][b:b + 1, :n]).to(dev),
               "bbox": torch.from_numpy(g["bbox"][b:b + 1, :n]).to(dev),
               "pixel_values": torch.from_numpy(g["pixel_values"][b:b + 1]).to(dev),
               "labels": torch.from_numpy(g["labels"][b:b + 1]).to(dev)}
        ids = m.generate(**enc, num_beams=1, max_length=int(g["max_length"]))
        ref = g["greedy_ids"][b]
        ref = ref[:1 + int(np.argmax(ref == shape.eos_token_id))]
        assert ids[0].cpu().tolist() == ref.tolist()
        ids5 = m.generate(**enc, num_beams=5, max_length=int(g["max_length"]))
        assert ids5[0].cpu().tolist()[:len(ref)] == ref.tolist()
    # batched forward (ref: curriculumTrainer.py:648-656): logits -> argmax accuracy path
    labels = torch.from_numpy(g["labels"]).to(dev)
    out = m(input_ids=torch.from_numpy(g["input_ids"]).to(dev), bbox=torch.from_numpy(g["bbox"]).to(dev),
            attention_mask=torch.from_numpy(g["attention_mask"]).to(dev),
            pixel_values=torch.from_numpy(g["pixel_values"]).to(dev), labels=labels,
            decoder_attention_mask=(labels != -100).long())
    assert out.logits.shape == g["logits"].shape
    assert np.abs(out.logits.cpu().numpy() - g["logits"]).max() < 0.015 * np.abs(g["logits"]).max() + 0.02
    assert abs(float(out.loss) - float(g["loss"])) < 2e-2
    live = g["labels"] != -100
    assert np.array_equal(out.logits.argmax(-1).cpu().numpy()[live], g["logits"].argmax(-1)[live])


def test_config_keeps_an_asymmetric_decoder_depth_and_tie_flag(tmp_path):
    """UdopConfig semantics (stock configuration_udop.py:43-71): num_decoder_layers defaults to num_layers only when absent;
    tie_word_embeddings is read from config.json (default True) and survives save / load."""
    c = MarkushgrapherConfig(num_layers=4, num_decoder_layers=2)
    assert c.num_layers == 4 and c.num_decoder_layers == 2 and c.to_shape().num_decoder_layers == 2
    assert MarkushgrapherConfig(num_layers=3).num_decoder_layers == 3
    assert MarkushgrapherConfig(num_layers=3, num_decoder_layers=None).num_decoder_layers == 3
    assert MarkushgrapherConfig().tie_word_embeddings is True and MarkushgrapherConfig().architecture_variant == "none"
    shape = synth.SHAPES["tiny"]
    m = MarkushgrapherForConditionalGeneration(MarkushgrapherConfig(**{**shape.to_dict(), "num_decoder_layers": 1, "tie_word_embeddings": False}))
    assert len(m.decoder.block._modules) == 1 and len(m.encoder.block._modules) == 2
    m.save_pretrained(str(tmp_path))
    c2 = MarkushgrapherConfig.from_pretrained(str(tmp_path))
    assert c2.num_decoder_layers == 1 and c2.tie_word_embeddings is False


def test_e1_branch_tensors_round_trip_and_missing_e1_is_loud(tmp_path):
    """`encoder.molscribe_*` tensors (the OCSR branch this package does not compute) are kept, saved and reloaded with their
    names, as the reference's helpers expect (ref: utils_model_loading.py:23,36; begin.py:151); a model that needs e1
    refuses to run without it unless told otherwise."""
    m, shape = tiny_model()
    assert not m.requires_e1()
    sd = {k: v for k, v in m.state_dict().items()}
    sd["encoder.molscribe_encoder.layers.0.blocks.0.attn.qkv.weight"] = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    sd["encoder.molscribe_projector.0.weight"] = torch.ones(2, 3)
    sd["encoder.molscribe_projector.0.bias"] = torch.zeros(2)
    sd["decoder.relative_bias.biases.0.relative_attention_bias.weight"] = torch.zeros(32, 2)
    missing, unexpected = m.load_state_dict(sd)
    assert not missing and not unexpected
    assert sorted(m.ignored_keys) == ["decoder.relative_bias.biases.0.relative_attention_bias.weight", "lm_head.weight"]
    assert m.requires_e1()
    assert sorted(m.encoder.molscribe_projector.state_dict().keys()) == ["0.bias", "0.weight"]
    assert torch.equal(m.encoder.molscribe_encoder.state_dict()["layers.0.blocks.0.attn.qkv.weight"], sd["encoder.molscribe_encoder.layers.0.blocks.0.attn.qkv.weight"])
    assert "encoder.molscribe_projector.0.weight" in m.state_dict()
    # begin.py:151: model.safe_load(model.encoder.molscribe_projector, projector_states)
    m.safe_load(m.encoder.molscribe_projector, {"0.weight": torch.full((2, 3), 2.0), "0.bias": torch.ones(2)})
    assert float(m.encoder.molscribe_projector.state_dict()["0.weight"][0, 0]) == 2.0
    m.save_pretrained(str(tmp_path))
    m2 = MarkushgrapherForConditionalGeneration.from_pretrained(str(tmp_path))
    assert torch.equal(m2.encoder.molscribe_projector.state_dict()["0.weight"], torch.full((2, 3), 2.0)) and m2.requires_e1()
    g = load_golden("g3_trained_tiny.npz")
    kw = dict(input_ids=torch.from_numpy(g["input_ids"]), bbox=torch.from_numpy(g["bbox"]), pixel_values=torch.from_numpy(g["pixel_values"]))
    with pytest.raises(RuntimeError, match="e1"):
        m.generate(**kw, max_length=8)
    with pytest.raises(RuntimeError, match="e1"):
        m(**kw, labels=torch.from_numpy(g["labels"]))
    plain, _ = tiny_model()
    plain.config.architecture_variant = "me-lf-stack-1"                 # ref: begin.py:120 with config/predict.yaml:12
    with pytest.raises(RuntimeError, match="me-lf-stack-1"):
        plain.generate(**kw, max_length=8)
    with pytest.warns(UserWarning, match="no MolScribe checkpoint"):
        plain.init_molscribe_weights()


def _e1_checkpoint_tensors():
    """A complete tiny OCSR branch under the names a MarkushGrapher-2 checkpoint is INFERRED to use: timm naming behind
    `encoder.molscribe_encoder.`, an nn.Sequential(Linear, GELU, Linear) behind `encoder.molscribe_projector.`."""
    from markushgrapher_amd import e1_shapes
    from tests.test_e1 import _to_timm
    s1 = e1_shapes.PRESETS["tiny"]
    sd1 = e1_shapes.recipe_state_dict(s1)
    out = {"encoder.molscribe_encoder." + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in _to_timm(sd1, prefix="transformer.").items()}
    for j, idx in enumerate((0, 2)):
        out[f"encoder.molscribe_projector.{idx}.weight"] = torch.from_numpy(sd1[f"proj.{j}.weight"])
        out[f"encoder.molscribe_projector.{idx}.bias"] = torch.from_numpy(sd1[f"proj.{j}.bias"])
    over = dict(image_size=s1.image_size, embed_dim=s1.embed_dim, depths=list(s1.depths), num_heads=list(s1.num_heads), window_size=s1.window_size)
    return s1, sd1, out, over


def test_complete_e1_branch_is_recognised_and_round_trips(tmp_path):
    """A checkpoint that carries the whole OCSR branch: the model knows it computes e1 itself, derives the branch geometry (projector
    sizes from the tensors, the rest from config.e1) and keeps everything through save_pretrained / from_pretrained."""
    m, shape = tiny_model()
    s1, sd1, extra, over = _e1_checkpoint_tensors()
    m.config.architecture_variant = "me-lf-stack-1"
    m.config.e1 = over
    sd = dict(m.state_dict())
    sd.update(extra)
    missing, unexpected = m.load_state_dict(sd)
    assert not missing and not unexpected
    assert m.requires_e1() and m.computes_e1()
    m._check_e1(None)                                   # nothing to complain about: the branch is complete
    shp, canon = m._e1_setup()
    assert shp.proj_dims == s1.proj_dims and shp.d_model == s1.d_model and shp.src_image_size == shape.image_size and shp.depths == s1.depths
    assert set(canon) == set(sd1) and all(np.array_equal(canon[k].float().numpy(), sd1[k]) for k in sd1)
    m.save_pretrained(str(tmp_path))
    m2 = MarkushgrapherForConditionalGeneration.from_pretrained(str(tmp_path))
    assert m2.config.architecture_variant == "me-lf-stack-1" and m2.config.e1["window_size"] == s1.window_size and m2.computes_e1()
    shp2, canon2 = m2._e1_setup()
    assert shp2 == shp and all(torch.equal(canon2[k], canon[k]) for k in canon)
    # MolScribe's own checkpoint file ({'encoder': {...}} behind `module.`) through init_molscribe_weights(path)
    ck = {"encoder": {"module." + k[len("encoder.molscribe_encoder."):]: v for k, v in extra.items() if k.startswith("encoder.molscribe_encoder.")}, "decoder": {}}
    torch.save(ck, str(tmp_path / "swin.pth"))
    m3, _ = tiny_model()
    m3.config.e1 = over
    m3.init_molscribe_weights(str(tmp_path / "swin.pth"))
    m3.safe_load(m3.encoder.molscribe_projector, {k[len("encoder.molscribe_projector."):]: v for k, v in extra.items() if "projector" in k})
    assert m3.computes_e1()
    shp3, canon3 = m3._e1_setup()
    assert shp3 == shp and all(torch.equal(canon3[k].float(), canon[k].float()) for k in canon)
    # a branch of the wrong width is refused with a message that names it
    bad = dict(extra)
    bad["encoder.molscribe_projector.2.weight"] = torch.zeros(32, 128)
    bad["encoder.molscribe_projector.2.bias"] = torch.zeros(32)
    m4, _ = tiny_model()
    m4.config.e1 = over
    sd4 = dict(m4.state_dict()); sd4.update(bad)
    m4.load_state_dict(sd4)
    with pytest.raises(RuntimeError, match="usable OCSR branch"):
        m4._check_e1(None)


@pytest.mark.gpu
def test_generate_on_a_me_lf_stack_1_model_without_e1_argument():
    """The reference's shipped configuration (config/predict.yaml:12 `architecture_variant: me-lf-stack-1`) through the HF surface:
    generate() / forward() / generate_queue() without `e1=` evaluate the OCSR branch themselves; ids equal those of the same call with
    the branch's tokens passed in, and the oracle chain SwinOracle.e1 -> Oracle(e1=...)."""
    from oracle.swin_oracle import SwinOracle
    from oracle.udop_oracle import Oracle
    m, shape = tiny_model()
    s1, sd1, extra, over = _e1_checkpoint_tensors()
    m.config.architecture_variant = "me-lf-stack-1"
    m.config.e1 = over
    sd = dict(m.state_dict()); sd.update(extra)
    m.load_state_dict(sd)
    dev = torch.device("cuda:0")
    m = m.to(dev)
    g = load_golden("g3_trained_tiny.npz")
    kw = dict(input_ids=torch.from_numpy(g["input_ids"]).to(dev), bbox=torch.from_numpy(g["bbox"]).to(dev),
              pixel_values=torch.from_numpy(g["pixel_values"]).to(dev), attention_mask=torch.from_numpy(g["attention_mask"]).to(dev))
    T = int(g["max_length"])
    ids = m.generate(**kw, max_length=T).cpu().numpy()
    e1 = m._eng()._e1_engine.encode(kw["pixel_values"])
    ids_pre = m.generate(**kw, max_length=T, e1=e1).cpu().numpy()
    assert np.array_equal(ids, ids_pre)
    with torch.no_grad():
        e1_ref = SwinOracle(s1, sd1).e1(g["pixel_values"]).numpy()
    assert np.abs(e1.cpu().numpy() - e1_ref).max() < 0.02 * np.abs(e1_ref).max() + 0.02
    w = {k: v.float().cpu().numpy() for k, v in m.state_dict().items() if not k.startswith("encoder.molscribe_")}
    ref = Oracle(shape, w).greedy(g["input_ids"], g["bbox"], g["pixel_values"], g["attention_mask"], max_length=T, e1=e1_ref)
    assert np.array_equal(ids[:, :ref.shape[1]], ref)
    out = m(**kw, labels=torch.from_numpy(g["labels"]).to(dev))
    assert torch.isfinite(out.logits).all() and out.loss is not None
    b5 = m.generate(**kw, num_beams=5, max_length=T).cpu().numpy()
    assert np.array_equal(b5, m.generate(**kw, num_beams=5, max_length=T, e1=e1).cpu().numpy())
    enc = [dict(input_ids=kw["input_ids"][i:i + 1], bbox=kw["bbox"][i:i + 1], pixel_values=kw["pixel_values"][i:i + 1]) for i in range(ids.shape[0])]
    rows = m.generate_queue(enc, max_length=T, slots=3, chunk=3)
    for i, r in enumerate(rows):
        one = m.generate(**enc[i], max_length=T).cpu().numpy()[0]
        assert np.array_equal(r.cpu().numpy(), one[:len(r)])


@pytest.mark.gpu
def test_generate_with_e1_tokens_and_opt_out_on_gpu():
    from oracle.udop_oracle import Oracle
    m, shape = tiny_model()
    m.config.architecture_variant = "me-lf-stack-1"
    m = m.to("cuda")
    g = load_golden("g3_trained_tiny.npz")
    dev = m.device
    kw = {k: torch.from_numpy(g[k]).to(dev) for k in ("input_ids", "bbox", "attention_mask", "pixel_values")}
    e1 = synth.round_bf16(synth.uniform_pm1("e1.tokens", (g["input_ids"].shape[0], 5, shape.d_model), 2) * np.float32(1.5))
    ids = m.generate(**kw, e1=torch.from_numpy(e1).to(dev), max_length=int(g["max_length"]))
    sd = dict(np.load(os.path.join(GOLDEN, "g3_weights.npz")))
    ref = Oracle(shape, sd).greedy(g["input_ids"], g["bbox"], g["pixel_values"], g["attention_mask"], max_length=int(g["max_length"]), e1=e1)
    assert ids.shape[1] == ref.shape[1] and (ids.cpu().numpy() == ref).mean() > 0.9      # trained fixture: margins >> noise
    m.config.allow_missing_e1 = True
    with pytest.warns(UserWarning, match="allow_missing_e1"):
        ids0 = m.generate(**kw, max_length=int(g["max_length"]))
    assert np.array_equal(ids0.cpu().numpy(), g["greedy_ids"])


@pytest.mark.gpu
def test_generate_queue_equals_the_per_image_loop():
    """generate_queue(encodings) against the reference's loop `for sample: model.generate(**encoding, num_beams=1, max_length=...)`
    (ref: utils/ocsr/utils_evaluation.py:140, 269-285) on per-sample encodings of DIFFERENT lengths: identical ids per image."""
    import json
    m, shape = tiny_model()
    m = m.to("cuda")
    dev = m.device
    with open(os.path.join(GOLDEN, "pipeline_host.json")) as f:
        pages = json.load(f)["pages"]
    g = load_golden("g3_trained_tiny.npz")
    encodings = []
    for k in range(7):
        p = pages[k % len(pages)]
        encodings.append({"input_ids": torch.tensor([p["input_ids"]]), "bbox": torch.tensor([p["bbox"]], dtype=torch.float32),
                          "pixel_values": torch.from_numpy(g["pixel_values"][k % g["pixel_values"].shape[0]][None])})
    loop = []
    for e in encodings:
        enc = {k: v.to(dev) for k, v in e.items()}
        loop.append(m.generate(**enc, num_beams=1, max_length=16)[0].cpu())
    got = m.generate_queue(encodings, max_length=16, slots=3, chunk=2)
    assert len(got) == len(loop)
    for a, b in zip(got, loop):
        b = b.tolist()
        n = b.index(shape.eos_token_id) + 1 if shape.eos_token_id in b else len(b)
        assert a.cpu().tolist() == b[:n]
    assert len({tuple(x.cpu().tolist()) for x in got}) > 1
    # the same queue cut over two execution contexts decoding at the same time (7 images, 3 slots each)
    for _ in range(2):
        got2 = m.generate_queue(encodings, max_length=16, slots=3, chunk=2, contexts=2)
        assert [x.cpu().tolist() for x in got2] == [x.cpu().tolist() for x in got]
    # the reference's shipped decode mode (config/predict.yaml beam_search: True): the same queue with beam search against the loop
    # `model.generate(**encoding, num_beams=3, max_length=...)` - the beam queue (image slots of 3 rows), one context and two
    loop_b = []
    for e in encodings:
        enc = {k: v.to(dev) for k, v in e.items()}
        loop_b.append(m.generate(**enc, num_beams=3, max_length=16)[0].cpu().tolist())
    for ctxs in (1, 2):
        got_b = m.generate_queue(encodings, max_length=16, slots=3, chunk=2, num_beams=3, contexts=ctxs)
        for a, b in zip(got_b, loop_b):
            a = a.cpu().tolist()
            assert a == b[:len(a)] and len(a) >= 2, (a, b)
    # and the per-image loop is unaffected afterwards (padding semantics restored)
    e0 = {k: v.to(dev) for k, v in encodings[0].items()}
    assert m.generate(**e0, num_beams=1, max_length=16)[0].cpu().tolist() == loop[0].tolist()


@pytest.mark.gpu
def test_in_flight_contexts_from_the_model():
    """model.in_flight(n): batches decoded at the same time on n contexts equal model.generate's ids."""
    m, shape = tiny_model()
    m = m.to("cuda")
    g = load_golden("g3_trained_tiny.npz")
    kw = {k: torch.from_numpy(g[k]).to(m.device) for k in ("input_ids", "bbox", "attention_mask", "pixel_values")}
    T = int(g["max_length"])
    want = m.generate(**kw, max_length=T).cpu().numpy()
    batches = [{k: v[i:i + 3] for k, v in kw.items()} for i in (0, 3, 1, 2)]

    def job(ctx, b):
        ids, _, _ = ctx.generate(b["input_ids"], b["bbox"], b["attention_mask"], b["pixel_values"], max_length=T)
        return ids.cpu().numpy()
    with m.in_flight(3) as fl:
        got = fl.map(job, batches)
    for b, o in zip((0, 3, 1, 2), got):
        w = want[b:b + 3]
        n = min(o.shape[1], w.shape[1])
        assert np.array_equal(o[:, :n], w[:, :n]) and np.all(w[:, n:] == shape.pad_token_id)
    assert np.array_equal(m.generate(**kw, max_length=T).cpu().numpy(), want)
