"""Mints tests/golden/swin_{tiny,w12,swin_b_384}.npz: outputs of STOCK transformers `SwinModel` - the importable upstream of the
MolScribe Swin-B encoder the reference loads into `model.encoder.molscribe_encoder` (ref: markushgrapher/core/common/begin.py:137-138,
utils/model/utils_model_loading.py:20-36, README.md:212-215) - on recipe weights and inputs (markushgrapher_amd/e1_shapes.py), after
asserting that oracle/swin_oracle.py reproduces them.  Only data is written.
    python tools/make_golden_swin.py [tiny] [w12] [swin_b_384]
Stored per fixture: the branch's input pixels are regenerated from the recipe (not stored); `features` = stock last_hidden_state
(full for the small presets, probe rows + per-row statistics for Swin-B), and the oracle's `e1` = projector(features) - that last
part and the resize in front are INFERRED pieces of the fork (parity unpinned), kept to detect drift of the build's own restatement.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from markushgrapher_amd.e1_shapes import PRESETS, recipe_state_dict, synth_pixels  # noqa: E402
from oracle.swin_oracle import SwinOracle  # noqa: E402


def stock_model(s, sd):
    from transformers import SwinConfig, SwinModel
    cfg = SwinConfig(image_size=s.image_size, patch_size=s.patch_size, num_channels=s.num_channels, embed_dim=s.embed_dim,
                     depths=list(s.depths), num_heads=list(s.num_heads), window_size=s.window_size, mlp_ratio=float(s.mlp_ratio),
                     qkv_bias=True, hidden_act="gelu", layer_norm_eps=s.layer_norm_eps, use_absolute_embeddings=False,
                     drop_path_rate=0.0)
    cfg._attn_implementation = "eager"
    m = SwinModel(cfg, add_pooling_layer=False).eval()
    own = {k[len("swin."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("swin.")}
    m.load_state_dict(own, strict=True)
    return m


def mint(name, B):
    s = PRESETS[name]
    sd = recipe_state_dict(s)
    src = synth_pixels(s, B)
    orc = SwinOracle(s, sd)
    with torch.no_grad():
        pix = orc.derive_input(src)
        t0 = time.time()
        m = stock_model(s, sd)
        nparam = sum(p.numel() for p in m.parameters())
        feats = m(pixel_values=pix).last_hidden_state
        t1 = time.time()
        of = orc.features(pix)
        t2 = time.time()
        e1 = orc.project(of)
        ofe = SwinOracle(s, sd, emulate_bf16=True).features(pix)
    err = float((of - feats).abs().max())
    print(f"[{name}] stock SwinModel: {nparam / 1e6:.2f} M parameters, output {tuple(feats.shape)}, {t1 - t0:.1f}s; oracle {t2 - t1:.1f}s; "
          f"oracle vs stock max {err:.2e} (|features| mean {float(feats.abs().mean()):.3f}, max {float(feats.abs().max()):.3f}); "
          f"bf16-emulating oracle vs stock max {float((ofe - feats).abs().max()):.3e}")
    assert tuple(feats.shape) == (B, s.out_tokens, s.out_dim)
    assert err < 2e-4, "oracle does not reproduce stock SwinModel"
    rows = np.arange(0, s.out_tokens, max(1, s.out_tokens // 16))[:16]
    out = dict(shape=np.array(name), B=B, n_params=nparam, probe_rows=rows, features_probe=feats[:, rows].numpy(),
               features_row_mean=feats.mean(dim=-1).numpy(), features_row_absmean=feats.abs().mean(dim=-1).numpy(),
               features_checksum=feats.double().sum(dim=(1, 2)).numpy(), features_absmax=np.float32(feats.abs().max()),
               e1_probe=e1[:, rows].numpy(), e1_checksum=e1.double().sum(dim=(1, 2)).numpy(), e1_absmax=np.float32(e1.abs().max()),
               input_checksum=pix.double().sum(dim=(1, 2, 3)).numpy(), flops_per_image=np.float64(SwinOracle.flops_per_image(s)),
               versions=np.array(f"transformers {__import__('transformers').__version__} torch {torch.__version__}"))
    if name != "swin_b_384":
        out["features"] = feats.numpy()
        out["e1"] = e1.numpy()
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz" if name.startswith("swin_") else f"swin_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    what = sys.argv[1:] or ["tiny", "w12", "swin_b_384"]
    if "tiny" in what:
        mint("tiny", 2)
    if "w12" in what:
        mint("w12", 2)
    if "swin_b_384" in what:
        mint("swin_b_384", 2)
