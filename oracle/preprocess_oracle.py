"""CPU ORACLE (test infrastructure only) for the page preprocessing contract of SURVEY.md §8 a12 / f-3:
   1024x1024 u8 RGB page --PIL LANCZOS--> 512x512 u8   (ref: markushgrapher/core/datasets/mdu_dataset.py:118)
   --> x/255 (float64 product cast to float32), (x - 0.5)/0.5 in float32, CHW
       (MarkushgrapherImageProcessor(apply_ocr=False, size=512), ref: core/common/begin.py:105-109; stock stand-in
        LayoutLMv3 image processor: rescale 1/255, mean = std = 0.5).
`lanczos_resize_u8` restates Pillow's 8-bit resampler (libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
ImagingResampleHorizontal_8bpc / Vertical_8bpc; PRECISION_BITS = 22, horizontal pass first).  Pinned bit-exactly against
Pillow itself in tests/test_preprocess.py (Pillow is importable in the build container and on the GPU box).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _lanczos(x):
    if -3.0 <= x < 3.0:
        def sinc(v):
            if v == 0.0:
                return 1.0
            v = v * math.pi
            return math.sin(v) / v
        return sinc(x) * sinc(x / 3.0)
    return 0.0


def lanczos_coeffs(in_size, out_size):
    """-> (xmin [out], count [out], kk int32 [out][ksize]) exactly as Pillow's precompute_coeffs + normalize_coeffs_8bpc."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 3.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / fscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = [_lanczos((x + lo - center + 0.5) * ss) for x in range(n)]
        ww = sum(w)          # Pillow accumulates in order
        tot = 0.0
        for v in w:
            tot += v
        if tot != 0.0:
            w = [v / tot for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        xmin[xx], cnt[xx] = lo, n
    return xmin, cnt, kk


def _resample_axis(img, out_size, axis):
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    xmin, cnt, kk = lanczos_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.int64)
    for xx in range(out_size):
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(cnt[xx]):
            acc += img[xmin[xx] + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def lanczos_resize_u8(img, out_h, out_w):
    """img u8 [H][W][C] -> u8 [out_h][out_w][C]; horizontal pass first, then vertical (Pillow's order)."""
    tmp = _resample_axis(img, out_w, 1) if img.shape[1] != out_w else img
    return _resample_axis(tmp, out_h, 0) if img.shape[0] != out_h else tmp


def normalize_u8(img_u8):
    """u8 HWC -> f32 CHW: float64 product with 1/255 cast to float32, then (x - 0.5) / 0.5 in float32."""
    x = (img_u8.astype(np.float64) * (1 / 255)).astype(np.float32)
    y = (x - np.float32(0.5)) / np.float32(0.5)
    return np.ascontiguousarray(y.transpose(2, 0, 1))


def preprocess_pages(pages_u8, out_size):
    return np.stack([normalize_u8(lanczos_resize_u8(p, out_size, out_size)) for p in pages_u8])
