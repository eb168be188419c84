"""Kernel-level parity through the C-ABI operator entry points (mgk_*), against plain numpy fp32 references of
the same op on bf16-rounded operands.  Parametrised over the backends of tests/backends.py: `hip` (real MI355X,
marked gpu) and `emu` (the same sources on the CPU SIMT emulator — index-math check, runs here)."""
import ctypes as C

import numpy as np
import pytest

from tests import pkutil as pk
from tests.backends import get_backend

BACKENDS = [pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)]
EPI_F32_STORE, EPI_F32_RESID, EPI_PK_RELU, EPI_PK = 0, 1, 2, 3
HF_PK_ROWS, HF_PK_T, HF_NATURAL, HF_STEP_Q, HF_STEP_KV = 1, 2, 3, 4, 5


def rnd(shape, seed, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32)


def ci(x):
    return C.c_int(int(x))


@pytest.mark.parametrize("be_name", BACKENDS)
def test_pack_weight(be_name):
    be = get_backend(be_name)
    N, K = 70, 128
    w = rnd((N, K), 1)
    src = be.buf(w)
    dst = be.zeros((96 * K,), np.uint16)
    assert be.lib.mgk_pack_weight(be.stream, be.p(src), 0, N, K, be.p(dst), 96) == 0
    ref = pk.pack_tiles(w, rows_pad=96)
    assert np.array_equal(dst.numpy(), ref)
    # bf16 source
    srcb = be.buf(pk.bf16_bits(w))
    dst2 = be.zeros((96 * K,), np.uint16)
    assert be.lib.mgk_pack_weight(be.stream, be.p(srcb), 1, N, K, be.p(dst2), 96) == 0
    assert np.array_equal(dst2.numpy(), ref)


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("mode,M,N,K", [(0, 160, 200, 128), (0, 128, 128, 64), (1, 32, 96, 256), (1, 20, 40, 64),
                                        (1, 70, 64, 128)])
def test_gemm_f32(be_name, mode, M, N, K):
    be = get_backend(be_name)
    x, w = rnd((M, K), 2), rnd((N, K), 3)
    bias = rnd((N,), 4)
    ref = pk.bf16_round(x) @ pk.bf16_round(w).T
    X, W = be.buf(pk.pack_tiles(x)), be.buf(pk.pack_tiles(w))
    out = be.zeros((M, N), np.float32)
    bb = be.buf(bias)
    assert be.lib.mgk_gemm(be.stream, mode, EPI_F32_STORE, be.p(X), be.p(W), M, N, K, be.p(out), N, be.p(bb), None) == 0
    np.testing.assert_allclose(out.numpy(), ref + bias, rtol=1e-4, atol=1e-4)
    # residual accumulate
    h0 = rnd((M, N), 5)
    h = be.buf(h0)
    assert be.lib.mgk_gemm(be.stream, mode, EPI_F32_RESID, be.p(X), be.p(W), M, N, K, be.p(h), N, None, None) == 0
    np.testing.assert_allclose(h.numpy(), h0 + ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("mode,M,N,K", [(0, 160, 192, 128), (1, 32, 128, 128), (1, 45, 64, 64)])
def test_gemm_packed_relu(be_name, mode, M, N, K):
    be = get_backend(be_name)
    x, w = rnd((M, K), 6), rnd((N, K), 7)
    ref = np.maximum(pk.bf16_round(x) @ pk.bf16_round(w).T, 0)
    X, W = be.buf(pk.pack_tiles(x)), be.buf(pk.pack_tiles(w))
    Mp = (M + 31) // 32 * 32
    out = be.zeros((Mp * N,), np.uint16)
    assert be.lib.mgk_gemm(be.stream, mode, EPI_PK_RELU, be.p(X), be.p(W), M, N, K, None, 0, None, be.p(out)) == 0
    got = pk.unpack_tiles(out.numpy(), M, N)
    np.testing.assert_allclose(got, ref, rtol=1.0 / 128, atol=1e-3)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_gemm_heads_flash_layout(be_name):
    """QKV projection of the encoder: Q,K packed rows, V packed transposed, per (b,h)."""
    be = get_backend(be_name)
    B, S, H, K = 2, 64, 2, 64
    inner = H * 64
    M, N = B * S, 3 * inner
    x, w = rnd((M, K), 8), rnd((N, K), 9, 0.2)
    ref = (pk.bf16_round(x) @ pk.bf16_round(w).T).reshape(B, S, 3, H, 64).transpose(2, 0, 3, 1, 4)  # [3][B][H][S][64]
    X, W = be.buf(pk.pack_tiles(x)), be.buf(pk.pack_tiles(w))
    q = be.zeros((B * H * S * 64,), np.uint16)
    k = be.zeros((B * H * S * 64,), np.uint16)
    v = be.zeros((B * H * S * 64,), np.uint16)
    rc = be.lib.mgk_gemm_heads(be.stream, 0, be.p(X), be.p(W), M, N, K, be.p(q), be.p(k), be.p(v),
                               HF_PK_ROWS, HF_PK_ROWS, HF_PK_T, H, S, S, None, 0)
    assert rc == 0
    np.testing.assert_allclose(pk.unpack_heads_rows(q.numpy(), B, H, S), ref[0], rtol=1 / 128, atol=1e-3)
    np.testing.assert_allclose(pk.unpack_heads_rows(k.numpy(), B, H, S), ref[1], rtol=1 / 128, atol=1e-3)
    np.testing.assert_allclose(pk.unpack_heads_t(v.numpy(), B, H, S), ref[2], rtol=1 / 128, atol=1e-3)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_gemm_heads_natural_compacted(be_name):
    """cross-K/V projection for decode: natural rows, compacted by row_map (masked tokens dropped)."""
    be = get_backend(be_name)
    B, S, H, K = 2, 64, 2, 64
    inner = H * 64
    M, N = B * S, 2 * inner
    x, w = rnd((M, K), 10), rnd((N, K), 11, 0.2)
    ref = (pk.bf16_round(x) @ pk.bf16_round(w).T).reshape(B, S, 2, H, 64)
    rm = np.full((B, S), -1, np.int32)
    for b in range(B):
        keep = np.arange(S) % (3 + b) != 0
        rm[b, keep] = np.arange(keep.sum())
    X, W = be.buf(pk.pack_tiles(x)), be.buf(pk.pack_tiles(w))
    kk = be.zeros((B, H, S, 64), np.uint16)
    vv = be.zeros((B, H, S, 64), np.uint16)
    rmb = be.buf(rm)
    rc = be.lib.mgk_gemm_heads(be.stream, 0, be.p(X), be.p(W), M, N, K, be.p(kk), be.p(vv), None,
                               HF_NATURAL, HF_NATURAL, 0, H, S, S, be.p(rmb), 0)
    assert rc == 0
    gk, gv = pk.bf16_to_f32(kk.numpy()), pk.bf16_to_f32(vv.numpy())
    for b in range(B):
        for s in range(S):
            r = rm[b, s]
            if r >= 0:
                np.testing.assert_allclose(gk[b, :, r], ref[b, s, 0], rtol=1 / 128, atol=1e-3)
                np.testing.assert_allclose(gv[b, :, r], ref[b, s, 1], rtol=1 / 128, atol=1e-3)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_gemm_heads_step(be_name):
    """decode-step QKV: q to [rows][H][64], k/v appended to the cache at `pos`."""
    be = get_backend(be_name)
    rows, H, K, T = 20, 2, 64, 16
    inner = H * 64
    x, w = rnd((rows, K), 12), rnd((3 * inner, K), 13, 0.2)
    ref = (pk.bf16_round(x) @ pk.bf16_round(w).T).reshape(rows, 3, H, 64)
    X, W = be.buf(pk.pack_tiles(x)), be.buf(pk.pack_tiles(w))
    q = be.zeros((rows, H, 64), np.uint16)
    kc = be.zeros((rows, H, T, 64), np.uint16)
    vc = be.zeros((rows, H, T, 64), np.uint16)
    pos = 5
    rc = be.lib.mgk_gemm_heads(be.stream, 1, be.p(X), be.p(W), rows, 3 * inner, K, be.p(q), be.p(kc), be.p(vc),
                               HF_STEP_Q, HF_STEP_KV, HF_STEP_KV, H, rows, T, None, pos)
    assert rc == 0
    np.testing.assert_allclose(pk.bf16_to_f32(q.numpy()), ref[:, 0], rtol=1 / 128, atol=1e-3)
    np.testing.assert_allclose(pk.bf16_to_f32(kc.numpy())[:, :, pos], ref[:, 1], rtol=1 / 128, atol=1e-3)
    np.testing.assert_allclose(pk.bf16_to_f32(vc.numpy())[:, :, pos], ref[:, 2], rtol=1 / 128, atol=1e-3)
    assert np.all(kc.numpy()[:, :, :pos] == 0) and np.all(kc.numpy()[:, :, pos + 1:] == 0)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_rmsnorm_pack(be_name):
    be = get_backend(be_name)
    M, d = 37, 128
    h, g = rnd((M, d), 14, 3.0), 1 + 0.2 * rnd((d,), 15)
    var = (h.astype(np.float32) ** 2).mean(-1, keepdims=True)
    ref = (g * (h * (1.0 / np.sqrt(var + 1e-6)))).astype(np.float32) * np.float32(0.125)
    hb, gb = be.buf(h), be.buf(g)
    xp = be.zeros((64 * d,), np.uint16)
    of = be.zeros((M, d), np.float32)
    assert be.lib.mgk_rmsnorm_pack(be.stream, be.p(hb), be.p(gb), be.p(xp), be.p(of), M, d, C.c_float(1e-6),
                                   C.c_float(0.125)) == 0
    np.testing.assert_allclose(of.numpy(), ref, rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(pk.unpack_tiles(xp.numpy(), M, d), ref, rtol=1 / 128, atol=1e-6)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_im2col_pack(be_name):
    be = get_backend(be_name)
    B, Cc, I, ps = 2, 3, 64, 16
    pix = rnd((B, Cc, I, I), 16)
    n = I // ps
    ref = pix.reshape(B, Cc, n, ps, n, ps).transpose(0, 2, 4, 1, 3, 5).reshape(B * n * n, Cc * ps * ps)
    pb = be.buf(pix)
    out = be.zeros((B * n * n * Cc * ps * ps,), np.uint16)
    assert be.lib.mgk_im2col_pack(be.stream, be.p(pb), be.p(out), B, Cc, I, ps) == 0
    np.testing.assert_array_equal(pk.unpack_tiles(out.numpy(), B * n * n, Cc * ps * ps), pk.bf16_round(ref))
