"""The drop-in hook (markushgrapher_amd/hf_shim.py): `from transformers.models.markushgrapher import ...` must resolve the way
the reference imports it (ref: markushgrapher/core/common/begin.py:7-13) and the model must be built and called the way the
reference builds and calls it (ref: begin.py:105-133, utils/ocsr/utils_evaluation.py:151-175,269-285)."""
import os

import numpy as np
import pytest
import torch

from markushgrapher_amd import synth
from tests.conftest import load_golden, GOLDEN


def _reference_style_load(tmp_path, variant="none"):
    import markushgrapher_amd.hf_shim  # noqa: F401  (before importing markushgrapher.*)
    # ref: begin.py:7-13
    from transformers.models.markushgrapher import (
        MarkushgrapherConfig,
        MarkushgrapherForConditionalGeneration,
        MarkushgrapherImageProcessor,
        MarkushgrapherProcessor,
        MarkushgrapherTokenizer,
    )
    shape = synth.SHAPES["tiny"]
    # a checkpoint directory as from_pretrained expects it
    src = MarkushgrapherForConditionalGeneration(MarkushgrapherConfig(**shape.to_dict()))
    sd = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "g3_weights.npz")).items()}
    src.load_state_dict(sd)
    src.save_pretrained(str(tmp_path))
    # ref: begin.py:105-121
    image_processor_no_ocr = MarkushgrapherImageProcessor(apply_ocr=False, size={"height": shape.image_size, "width": shape.image_size})
    config = MarkushgrapherConfig.from_pretrained(str(tmp_path))
    config.image_size = shape.image_size
    config.architecture_variant = variant
    config.output_attentions = True
    return (MarkushgrapherForConditionalGeneration, MarkushgrapherProcessor, MarkushgrapherTokenizer, image_processor_no_ocr,
            config, shape)


def test_shim_resolves_the_five_names_and_builds_the_model(tmp_path):
    import transformers
    Model, Processor, Tokenizer, ip, config, shape = _reference_style_load(tmp_path)
    import transformers.models.markushgrapher as mod
    assert mod.MarkushgrapherForConditionalGeneration is Model
    assert issubclass(Tokenizer, transformers.UdopTokenizer) or Tokenizer is transformers.UdopTokenizer
    assert Processor is transformers.UdopProcessor
    # the image processor the reference builds (apply_ocr=False, size=512^2 there) rescales by 1/255 and normalises with 0.5/0.5
    px = ip(images=[np.full((shape.image_size, shape.image_size, 3), 255, np.uint8)], return_tensors="pt")["pixel_values"]
    assert tuple(px.shape) == (1, 3, shape.image_size, shape.image_size) and float(px.max()) == 1.0
    # ref: begin.py:128-133 (both constructors)
    m1 = Model(config)
    m2 = Model.from_pretrained(str(tmp_path), config=config)
    assert m2.config.output_attentions is True and m2.config.image_size == shape.image_size
    for k, v in m2.state_dict().items():
        assert k in m1.state_dict() and v.shape == m1.state_dict()[k].shape
    # ref: begin.py:166 model.safe_load(model.decoder, decoder_states)
    m1.safe_load(m1.decoder, m2.decoder.state_dict())
    assert torch.equal(m1.decoder.state_dict()["block.0.layer.0.SelfAttention.q.weight"],
                       m2.decoder.state_dict()["block.0.layer.0.SelfAttention.q.weight"])


@pytest.mark.gpu
def test_shim_model_generates_like_get_smiles_metrics_calls_it(tmp_path):
    Model, _, _, _, config, shape = _reference_style_load(tmp_path)
    device = torch.device("cuda")
    model = Model.from_pretrained(str(tmp_path), config=config).to(device)       # begin.py:130-133
    g = load_golden("g3_trained_tiny.npz")
    for b in range(2):
        n = int(g["attention_mask"][b].sum())
        # ref: utils_evaluation.py:151-175: per-sample tensors on the device, attention_mask deleted, labels left in the kwargs
        encoding = {"input_ids": torch.from_numpy(g["input_ids"][b:b + 1, :n]).to(device),
                    "bbox": torch.from_numpy(g["bbox"][b:b + 1, :n]).to(device),
                    "pixel_values": torch.from_numpy(g["pixel_values"][b:b + 1]).to(device),
                    "labels": torch.from_numpy(g["labels"][b:b + 1]).to(device)}
        # ref: utils_evaluation.py:269-285
        if hasattr(model, "module"):
            predictions = model.module.generate(**encoding, num_beams=5, max_length=512)
        else:
            predictions = model.generate(**encoding, num_beams=5, max_length=512)
        ref = g["beam_ids"][b]
        ref = ref[:1 + int(np.argmax(ref == shape.eos_token_id))]
        assert predictions[0].cpu().tolist()[:len(ref)] == ref.tolist()
