"""Page preprocessing ("next" row f-3): oracle vs Pillow / the stock image processor (CPU), HIP kernel vs oracle."""
import ctypes as C

import numpy as np
import pytest

from oracle import preprocess_oracle as po
from tests.backends import get_backend

BACKENDS = [pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)]


def test_oracle_matches_pillow_lanczos_bit_exactly():
    from PIL import Image
    rs = np.random.RandomState(1)
    for (h, w, oh, ow) in [(64, 64, 32, 32), (128, 96, 64, 48), (100, 60, 50, 30), (90, 90, 30, 30), (256, 256, 128, 128)]:
        a = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        a[: h // 4] = 255                                  # white page regions, as in the chemical pages
        ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.LANCZOS))
        assert np.array_equal(po.lanczos_resize_u8(a, oh, ow), ref)


def test_oracle_normalisation_matches_stock_image_processor():
    transformers = pytest.importorskip("transformers")
    from PIL import Image
    rs = np.random.RandomState(2)
    a = rs.randint(0, 256, (32, 32, 3)).astype(np.uint8)
    p = transformers.LayoutLMv3ImageProcessor(apply_ocr=False, size={"height": 32, "width": 32})
    ref = p(images=Image.fromarray(a), return_tensors="np")["pixel_values"][0]
    assert np.array_equal(po.normalize_u8(a), ref)


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("B,Hs,Ws,out", [(2, 64, 64, 32), (1, 96, 128, 32), (3, 90, 90, 30)])
def test_preprocess_kernel_bit_exact(be_name, B, Hs, Ws, out):
    be = get_backend(be_name)
    rs = np.random.RandomState(3)
    pages = rs.randint(0, 256, (B, Hs, Ws, 3)).astype(np.uint8)
    pages[:, :5] = 255
    ref = po.preprocess_pages(pages, out)
    be.lib.mg_preprocess_scratch_bytes.restype = C.c_size_t
    nb = be.lib.mg_preprocess_scratch_bytes(B, Hs, Ws, out)
    scratch = be.zeros((nb,), np.uint8)
    pv = be.zeros((B, 3, out, out), np.float32)
    rc = be.lib.mg_preprocess_pages(be.stream, be.p(be.buf(pages)), B, Hs, Ws, out, be.p(pv), be.p(scratch), C.c_size_t(nb))
    assert rc == 0
    assert np.array_equal(pv.numpy(), ref)


@pytest.mark.gpu
def test_preprocess_full_size_page_against_pillow():
    """1024x1024 -> 512x512 (the reference's sizes): device result == Pillow LANCZOS + stock normalisation, bit for bit."""
    from PIL import Image
    from markushgrapher_amd import synth
    be = get_backend("hip")
    pages = synth.synth_pages_u8(2, 1024, seed=9)
    ref = np.stack([po.normalize_u8(np.asarray(Image.fromarray(p).resize((512, 512), Image.LANCZOS))) for p in pages])
    be.lib.mg_preprocess_scratch_bytes.restype = C.c_size_t
    nb = be.lib.mg_preprocess_scratch_bytes(2, 1024, 1024, 512)
    scratch = be.zeros((nb,), np.uint8)
    pv = be.zeros((2, 3, 512, 512), np.float32)
    assert be.lib.mg_preprocess_pages(be.stream, be.p(be.buf(pages)), 2, 1024, 1024, 512, be.p(pv), be.p(scratch), C.c_size_t(nb)) == 0
    assert np.array_equal(pv.numpy(), ref)
