"""Build-container-only helper: construct stock transformers' UdopForConditionalGeneration (the importable
upstream of the reference's un-vendored fork) for a ModelShape and load recipe weights.
Never imported by the product, tests marked gpu, bench.py or smoke(); never travels to the GPU box."""
import torch

from markushgrapher_amd import synth


def stock_config(shape):
    from transformers import UdopConfig
    cfg = UdopConfig(vocab_size=shape.vocab_size, d_model=shape.d_model, d_kv=shape.d_kv, d_ff=shape.d_ff,
                     num_layers=shape.num_layers, num_decoder_layers=shape.num_decoder_layers,
                     num_heads=shape.num_heads, max_2d_position_embeddings=shape.max_2d_position_embeddings,
                     image_size=shape.image_size, patch_size=shape.patch_size, dropout_rate=0.0,
                     decoder_start_token_id=shape.decoder_start_token_id)
    cfg._attn_implementation = "eager"   # closest to the fork's 4.34 order of operations
    return cfg


def stock_model(shape, sd):
    from transformers import UdopForConditionalGeneration
    m = UdopForConditionalGeneration(stock_config(shape)).eval()
    full = {k: torch.from_numpy(v) for k, v in sd.items()}
    for a, c in synth.tied_aliases(shape).items():
        full[a] = full[c]
    missing, unexpected = m.load_state_dict(full, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("decoder.embed_patches") or k.startswith("decoder.relative_bias") for k in missing), missing
    m.tie_weights()
    return m
