"""Mints tests/golden/ocr_tiny.npz and ocr_smoldocling.npz: outputs of STOCK transformers `Idefics3ForConditionalGeneration`
(the class the reference's ChemicalOCR loads, markushgrapher/ocr/chemical_ocr.py:76-84) on recipe weights and inputs
(markushgrapher_amd/ocr_shapes.py), after asserting that oracle/ocr_oracle.py reproduces them.  Only data is written.
    python tools/make_golden_ocr.py [tiny] [tiny2] [tiny3] [ragged] [smoldocling]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from markushgrapher_amd.ocr_shapes import PRESETS, recipe_state_dict, synth_inputs  # noqa: E402
from oracle.ocr_oracle import OcrOracle  # noqa: E402


def stock_model(s, sd):
    from transformers import Idefics3Config, Idefics3ForConditionalGeneration
    cfg = Idefics3Config(
        vision_config=dict(hidden_size=s.v_hidden, intermediate_size=s.v_inter, num_hidden_layers=s.v_layers,
                           num_attention_heads=s.v_heads, image_size=s.image_size, patch_size=s.patch_size, num_channels=3,
                           hidden_act="gelu_pytorch_tanh", layer_norm_eps=s.v_eps),
        text_config=dict(model_type="llama", hidden_size=s.t_hidden, intermediate_size=s.t_inter, num_hidden_layers=s.t_layers,
                         num_attention_heads=s.t_heads, num_key_value_heads=s.t_kv_heads, vocab_size=s.vocab,
                         rms_norm_eps=s.rms_eps, max_position_embeddings=8192, rope_theta=s.rope_theta,
                         tie_word_embeddings=s.tie_word_embeddings, pad_token_id=s.pad_token_id, bos_token_id=0,
                         eos_token_id=s.eos_token_id, head_dim=64),
        scale_factor=s.scale_factor, image_token_id=s.image_token_id, pad_token_id=s.pad_token_id,
        tie_word_embeddings=s.tie_word_embeddings)
    cfg._attn_implementation = "eager"
    m = Idefics3ForConditionalGeneration(cfg).eval()
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m


def mint(name, B, new_tokens, gain, eos_from_step=None, n_img=1, tag=None, masked=False):
    import dataclasses
    s = PRESETS[name]
    sd = recipe_state_dict(s, gain=gain)
    ids, pix = synth_inputs(s, B, n_img=n_img)
    t0 = time.time()
    m = stock_model(s, sd)
    tid, tpix = torch.from_numpy(ids), torch.from_numpy(pix)
    pam = torch.ones(B, n_img, s.image_size, s.image_size, dtype=torch.bool)
    if masked:       # non-square pages: the processor pads the frame and masks the padding (rows at the bottom / columns at the right)
        from markushgrapher_amd.ocr_shapes import synth_pixel_mask
        pam = torch.from_numpy(synth_pixel_mask(s, B, n_img))
        tpix = torch.where(pam[:, :, None], tpix, torch.zeros_like(tpix))
        pix = tpix.numpy()
    kw = dict(input_ids=tid, attention_mask=torch.ones_like(tid), pixel_values=tpix, pixel_attention_mask=pam)
    with torch.no_grad():
        feats = m.model.get_image_features(tpix, kw["pixel_attention_mask"], return_dict=True).pooler_output
        logits = m(**kw).logits
        gen = m.generate(**kw, max_new_tokens=new_tokens, do_sample=False, output_scores=True, return_dict_in_generate=True)
    new = gen.sequences[:, ids.shape[1]:]
    scores = torch.stack(gen.scores, dim=1)                      # [B][n][V] (raw logits: no processors are active)
    if eos_from_step is not None:
        # a second run whose EOS id is what row 0 emitted at that step: rows end at different steps, the rest is padded
        s = dataclasses.replace(s, eos_token_id=int(new[0, eos_from_step]))
        m.generation_config.eos_token_id = s.eos_token_id
        with torch.no_grad():
            gen = m.generate(**kw, max_new_tokens=new_tokens, do_sample=False, output_scores=True, return_dict_in_generate=True,
                             eos_token_id=s.eos_token_id)
        new = gen.sequences[:, ids.shape[1]:]
        scores = torch.stack(gen.scores, dim=1)
    print(f"[{name}] stock ran in {time.time() - t0:.1f}s; new ids row 0: {new[0].tolist()}")
    orc = OcrOracle(s, sd)
    with torch.no_grad():
        opam = pam.numpy() if masked else None
        of = orc.image_features(pix, opam)
        ol = orc.forward(ids, pix, opam)
        on, osc = orc.generate(ids, pix, new_tokens, return_logits=True, pixel_attention_mask=opam)
    e_f = float((of - feats).abs().max()); e_l = float((ol - logits).abs().max())
    n = min(on.shape[1], new.shape[1])
    same = bool((on[:, :n] == new[:, :n]).all()) and on.shape[1] == new.shape[1]
    e_s = float((osc[:, :n] - scores[:, :n]).abs().max())
    print(f"[{name}] oracle vs stock: image features {e_f:.2e}, teacher-forced logits {e_l:.2e} (max |logit| {float(logits.abs().max()):.2f}), "
          f"step logits {e_s:.2e}, ids equal {same}")
    assert e_f < 2e-4 and e_l < 2e-4 * max(1.0, float(logits.abs().max())) and same, "oracle does not reproduce stock"
    top = torch.topk(scores, 8, dim=-1)
    ltop = torch.topk(logits, 8, dim=-1)
    srt = torch.sort(scores, dim=-1, descending=True).values
    out = dict(shape=np.array(name), n_img=n_img, masked=int(masked), B=B, new_tokens=new_tokens, gain=np.float32(gain), eos_token_id=s.eos_token_id,
               input_ids=ids, new_ids=new.numpy(), step_top8_val=top.values.numpy(), step_top8_idx=top.indices.numpy(),
               step_margin=(srt[..., 0] - srt[..., 1]).numpy(), logits_top8_val=ltop.values.numpy(), logits_top8_idx=ltop.indices.numpy(),
               logits_absmax=np.float32(logits.abs().max()), feats_probe=feats[:, ::max(1, feats.shape[1] // 4)].numpy(),
               feats_checksum=feats.double().sum(dim=(1, 2)).numpy(), feats_abs_mean=np.float32(feats.abs().mean()),
               versions=np.array(f"transformers {__import__('transformers').__version__} torch {torch.__version__}"))
    if name == "tiny":
        out["logits"] = logits.numpy()
        out["feats"] = feats.numpy()
    path = os.path.join(ROOT, "tests", "golden", f"ocr_{tag or name}.npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} {os.path.getsize(path)} bytes; margins min {float(out['step_margin'].min()):.4f} median {float(np.median(out['step_margin'])):.4f}")


def mint_ragged(B=3, new_tokens=8, gain=0.7, drops=(0, 3, 5)):
    """ocr_tiny_ragged.npz: prompts of DIFFERENT lengths as one left-padded batch (what the Idefics3 processor returns for them:
    padding_side = left) through stock generate(); the oracle restates it as every row alone (generate_padded)."""
    s = PRESETS["tiny"]
    sd = recipe_state_dict(s, gain=gain)
    ids, pix = synth_inputs(s, B)
    L = ids.shape[1]
    padded, mask = np.full_like(ids, s.pad_token_id), np.zeros_like(ids)
    for b in range(B):
        d = drops[b]
        row = ids[b, :L - d]                      # the row's last d text tokens left out (every row keeps its <image> block)
        padded[b, d:] = row
        mask[b, d:] = 1
    m = stock_model(s, sd)
    m.generation_config.pad_token_id = s.pad_token_id
    tid, tam, tpix = torch.from_numpy(padded), torch.from_numpy(mask), torch.from_numpy(pix)
    pam = torch.ones(B, 1, s.image_size, s.image_size, dtype=torch.bool)
    with torch.no_grad():
        gen = m.generate(input_ids=tid, attention_mask=tam, pixel_values=tpix, pixel_attention_mask=pam, max_new_tokens=new_tokens, do_sample=False,
                         output_scores=True, return_dict_in_generate=True)
    new = gen.sequences[:, L:]
    scores = torch.stack(gen.scores, dim=1)
    orc = OcrOracle(s, sd)
    with torch.no_grad():
        on, osc = orc.generate_padded(padded, mask, pix, new_tokens, return_logits=True)
    n = min(on.shape[1], new.shape[1])
    same = bool((on[:, :n] == new[:, :n]).all()) and on.shape[1] == new.shape[1]
    e_s = float((osc[:, :n] - scores[:, :n]).abs().max())
    print(f"[ragged] stock left-padded batch vs the oracle's row-alone restatement: step logits {e_s:.2e} (max |logit| {float(scores.abs().max()):.2f}), ids equal {same}; "
          f"new ids {new.tolist()}")
    assert same and e_s < 2e-4 * max(1.0, float(scores.abs().max())), "oracle does not reproduce stock on the left-padded batch"
    top = torch.topk(scores, 8, dim=-1)
    srt = torch.sort(scores, dim=-1, descending=True).values
    path = os.path.join(ROOT, "tests", "golden", "ocr_tiny_ragged.npz")
    np.savez_compressed(path, shape=np.array("tiny"), B=B, new_tokens=new_tokens, gain=np.float32(gain), drops=np.array(drops), input_ids=padded,
                        attention_mask=mask, new_ids=new.numpy(), step_top8_val=top.values.numpy(), step_top8_idx=top.indices.numpy(),
                        step_margin=(srt[..., 0] - srt[..., 1]).numpy(),
                        versions=np.array(f"transformers {__import__('transformers').__version__} torch {torch.__version__}"))
    print(f"[ragged] wrote {path} {os.path.getsize(path)} bytes; margins min {float((srt[..., 0] - srt[..., 1]).min()):.4f}")


if __name__ == "__main__":
    what = sys.argv[1:] or ["tiny", "smoldocling"]
    if "tiny" in what:
        mint("tiny", B=3, new_tokens=12, gain=0.7, eos_from_step=5)
    if "tiny2" in what or not sys.argv[1:]:
        mint("tiny", B=2, new_tokens=6, gain=0.7, n_img=2, tag="tiny2")      # two frames per page (a page split by the processor)
    if "tiny3" in what or not sys.argv[1:]:
        mint("tiny", B=3, new_tokens=6, gain=0.7, tag="tiny3", masked=True)   # partially masked frames (non-square pages)
    if "ragged" in what or not sys.argv[1:]:
        mint_ragged()
    if "smoldocling" in what:
        mint("smoldocling", B=2, new_tokens=8, gain=1.0)
