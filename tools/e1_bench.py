"""Times the OCSR vision branch (mg_e1_encode) at the Swin-B geometry: ms per batch and achieved flop rate.
    python tools/e1_bench.py [B] [reps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from markushgrapher_amd.e1 import E1Engine  # noqa: E402
from markushgrapher_amd.e1_shapes import PRESETS, recipe_state_dict, synth_pixels  # noqa: E402


def flops_per_image(s):
    g = s.grid
    f = 2.0 * g * g * s.embed_dim * s.num_channels * s.patch_size ** 2
    for i in range(s.n_stages):
        R, C = s.stage_res(i), s.stage_dim(i)
        w = s.stage_window(i)
        f += s.depths[i] * (R * R * 2.0 * C * C * (4 + 2 * s.mlp_ratio) + R * R * 4.0 * w * w * C)
        if i + 1 < s.n_stages:
            f += (R * R / 4) * 2.0 * 4 * C * 2 * C
    dims = (s.out_dim,) + tuple(s.proj_dims) + (s.d_model,)
    for a, b in zip(dims[:-1], dims[1:]):
        f += s.out_tokens * 2.0 * a * b
    return f


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    s = PRESETS["swin_b_384"]
    eng = E1Engine(s).load_state_dict(recipe_state_dict(s))
    pix = torch.from_numpy(np.concatenate([synth_pixels(s, 2, seed=i) for i in range((B + 1) // 2)])[:B]).cuda()
    for _ in range(2):
        eng.encode(pix)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        eng.encode(pix)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / reps * 1e3
    fl = flops_per_image(s) * B
    print(f"e1 branch, Swin-B 384 px, B = {B}: {ms:.2f} ms per batch, {fl / 1e12:.2f} TFLOP -> {fl / ms / 1e12:.3f} PFLOP/s "
          f"({flops_per_image(s) / 1e9:.1f} GFLOP per image)")
