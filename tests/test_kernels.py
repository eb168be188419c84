"""Kernel-level parity through the C-ABI operator entry points (mgk_*), against plain numpy fp32 references of
the same op on bf16-rounded operands.  Parametrised over the backends of tests/backends.py: `hip` (real MI355X,
marked gpu) and `emu` (the same sources on the CPU SIMT emulator — index-math check, runs here)."""
import ctypes as C

import numpy as np
import pytest

from tests import pkutil as pk
from tests.backends import get_backend

BACKENDS = [pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)]


@pytest.fixture(autouse=True)
def _default_gemm_variant():
    """Tests that select a GEMM tile-kernel variant leave the library on its default afterwards, whatever happened in between."""
    yield
    from tests import backends as _b
    for be in _b._cache.values():
        be.lib.mgk_gemm_set_variant(3)

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
                                        (1, 70, 64, 128), (0, 300, 200, 192), (0, 512, 128, 64), (0, 288, 136, 320)])
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
@pytest.mark.parametrize("mode,M,N,K", [(0, 160, 192, 128), (1, 32, 128, 128), (1, 45, 64, 64), (0, 320, 256, 192)])
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
@pytest.mark.parametrize("variant", [2, 4, 5, 6])
@pytest.mark.parametrize("M,N,K", [(320, 256, 64), (300, 520, 128), (512, 384, 192), (700, 256, 128), (1100, 768, 256)])
def test_gemm_256x256_tile_kernel(be_name, M, N, K, variant):
    """variants 2 / 4: the 256x256x64 and 320x256x64 two-stage kernels (ragged edges in M and N, several K-steps), all
    epilogue families.  Variants 5 / 6: the persistent ping-pong kernel (k_gemm_pp.hip) with 256- / 320-row tiles for K % 128 == 0
    (other K fall back to the two-stage kernel); (1100, 768, 256) gives every workgroup of the emulator's 8 several tiles."""
    if be_name == "emu" and M > 1000 and variant in (2, 4):
        pytest.skip("large case: persistent kernel only on the emulator")
    be = get_backend(be_name)
    be.lib.mgk_gemm_set_variant(variant)
    try:
        x, w = rnd((M, K), 31), rnd((N, K), 32, 0.3)
        bias = rnd((N,), 33)
        ref = pk.bf16_round(x) @ pk.bf16_round(w).T
        X, W = be.buf(pk.pack_tiles(x)), be.buf(pk.pack_tiles(w))
        out = be.zeros((M, N), np.float32)
        bb = be.buf(bias)
        assert be.lib.mgk_gemm(be.stream, 0, EPI_F32_STORE, be.p(X), be.p(W), M, N, K, be.p(out), N, be.p(bb), None) == 0
        np.testing.assert_allclose(out.numpy(), ref + bias, rtol=1e-4, atol=1e-4)
        h0 = rnd((M, N), 34)
        h = be.buf(h0)
        assert be.lib.mgk_gemm(be.stream, 0, EPI_F32_RESID, be.p(X), be.p(W), M, N, K, be.p(h), N, None, None) == 0
        np.testing.assert_allclose(h.numpy(), h0 + ref, rtol=1e-4, atol=1e-4)
        Mp = (M + 31) // 32 * 32
        if N % 16 == 0:
            outp = be.zeros((Mp * N,), np.uint16)
            assert be.lib.mgk_gemm(be.stream, 0, EPI_PK_RELU, be.p(X), be.p(W), M, N, K, None, 0, None, be.p(outp)) == 0
            np.testing.assert_allclose(pk.unpack_tiles(outp.numpy(), M, N), np.maximum(ref, 0), rtol=1.0 / 128, atol=1e-3)
            if M >= 320:      # GELU(tanh) epilogue (ChemicalOCR vision MLP), large-M tile kernels only
                outg = be.zeros((Mp * N,), np.uint16)
                assert be.lib.mgk_gemm(be.stream, 0, 7, be.p(X), be.p(W), M, N, K, None, 0, None, be.p(outg)) == 0
                gl = 0.5 * ref * (1.0 + np.tanh(0.7978845608028654 * (ref + 0.044715 * ref ** 3)))
                np.testing.assert_allclose(pk.unpack_tiles(outg.numpy(), M, N), gl, rtol=1.0 / 128, atol=2e-3)
        if M % 64 == 0 and N % 384 == 0:      # per-head epilogue: Q, K packed rows, V packed transposed
            S, H = 64, N // 192
            B = M // S
            q, k, v = (be.zeros((B * H * S * 64,), np.uint16) for _ in range(3))
            rc = be.lib.mgk_gemm_heads(be.stream, 0, be.p(X), be.p(W), M, N, K, be.p(q), be.p(k), be.p(v),
                                       HF_PK_ROWS, HF_PK_ROWS, HF_PK_T, H, S, S, None, 0)
            assert rc == 0
            r5 = ref.reshape(B, S, 3, H, 64).transpose(2, 0, 3, 1, 4)
            np.testing.assert_allclose(pk.unpack_heads_rows(q.numpy(), B, H, S), r5[0], rtol=1 / 128, atol=1e-3)
            np.testing.assert_allclose(pk.unpack_heads_rows(k.numpy(), B, H, S), r5[1], rtol=1 / 128, atol=1e-3)
            np.testing.assert_allclose(pk.unpack_heads_t(v.numpy(), B, H, S), r5[2], rtol=1 / 128, atol=1e-3)
    finally:
        be.lib.mgk_gemm_set_variant(3)


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("Bn", [2, 5])      # M = 128 -> 128x128 kernel, M = 320 -> 256x128 three-stage kernel
def test_gemm_heads_flash_layout(be_name, Bn):
    """QKV projection of the encoder: Q,K packed rows, V packed transposed, per (b,h)."""
    be = get_backend(be_name)
    B, S, H, K = Bn, 64, 2, 64
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


# ---------------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------------
def pack_heads_rows(x):
    """fp32 [B][H][S][64] -> HF_PK_ROWS bits"""
    B, H, S, D = x.shape
    t = pk.bf16_bits(x).reshape(B, H, S // 32, 32, 4, 2, 8).transpose(0, 1, 2, 4, 5, 3, 6)
    return np.ascontiguousarray(t).reshape(-1)


def pack_heads_t(x):
    """fp32 [B][H][S][64] -> HF_PK_T bits  [b][h][dt][kt][half][dim32][8 tok]"""
    B, H, S, D = x.shape
    t = pk.bf16_bits(x).reshape(B, H, S // 16, 2, 8, 2, 32).transpose(0, 1, 5, 2, 3, 6, 4)
    return np.ascontiguousarray(t).reshape(-1)


def enc_tables(w1, wh, wv):
    import torch
    from oracle.udop_oracle import relative_position_bucket as rpb
    b1 = rpb(torch.arange(-128, 129), True, 32, 128).numpy()
    bhv = rpb(torch.arange(-100, 101), True, 32, 100).numpy()
    return w1[b1].astype(np.float32), wh[bhv].astype(np.float32), wv[bhv].astype(np.float32)


def softmax_ref(scores):
    m = scores.max(-1, keepdims=True)
    e = np.exp(scores - m)
    return e / e.sum(-1, keepdims=True)


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("S,S_cap,B,H", [(23, 64, 2, 2), (150, 192, 2, 2), (700, 704, 2, 2), (1085, 1088, 1, 16), (1150, 1280, 8, 2)])
def test_attention_encoder_bias(be_name, S, S_cap, B, H):
    """Sizes: one stage; a few stages; far tiles on both sides of the diagonal and several query blocks; the benchmark's
    shape (17 stages, 16 heads: the software pipeline at full depth); 8 images (XCD-aware workgroup mapping)."""
    import torch
    from oracle.udop_oracle import relative_position_bucket as rpb
    be = get_backend(be_name)
    if be_name == "emu" and B * H * S > 20000:
        pytest.skip("emulator: index math is covered by the smaller sizes")
    q, k, v = [pk.bf16_round(rnd((B, H, S_cap, 64), 20 + i, 0.5)) for i in range(3)]
    rs = np.random.RandomState(5)
    cx, cy = rs.rand(B, S_cap), rs.rand(B, S_cap)
    cx[:, 3] = cx[:, 4]   # zero horizontal distance
    cx = (np.round(cx * 64) / 64.0)   # exact multiples: distances land exactly on integer *100 boundaries sometimes
    mask = np.ones((B, S_cap), np.uint8)
    mask[B - 1, 5:9] = 0
    mask[:, S:] = 0
    w1, wh, wv = [rnd((32, H), 30 + i) for i in range(3)]
    t1, th, tv = enc_tables(w1, wh, wv)
    # reference (stock:904-953 semantics)
    pos = np.arange(S)
    b1 = rpb(torch.from_numpy(pos[None, :] - pos[:, None]), True, 32, 128).numpy()
    ref = np.zeros((B, H, S, 64), np.float32)
    for b in range(B):
        dx = ((torch.from_numpy(cx[b, None, :S] - cx[b, :S, None]).float() * 100).to(torch.long))
        dy = ((torch.from_numpy(cy[b, None, :S] - cy[b, :S, None]).float() * 100).to(torch.long))
        bh, bv = rpb(dx, True, 32, 100).numpy(), rpb(dy, True, 32, 100).numpy()
        for h in range(H):
            sc = q[b, h, :S] @ k[b, h, :S].T + w1[b1, h] + wh[bh, h] + wv[bv, h]
            sc = np.where(mask[b, None, :S] != 0, sc, -1e30)
            ref[b, h] = softmax_ref(sc) @ v[b, h, :S]
    Q, K, V = be.buf(pack_heads_rows(q)), be.buf(pack_heads_rows(k)), be.buf(pack_heads_t(v))
    ctx = be.zeros((B * S_cap * H * 64,), np.uint16)
    bk1 = rpb(torch.arange(-128, 129), True, 32, 128).numpy().astype(np.int32)
    bkhv = rpb(torch.arange(-100, 101), True, 32, 100).numpy().astype(np.int32)
    bidx = be.zeros((B * S_cap * S_cap,), np.uint16)
    rc = be.lib.mgk_attention(be.stream, 0, be.p(Q), be.p(K), be.p(V), be.p(ctx), B, H, S, S, S_cap, S_cap,
                              be.p(be.buf(mask)), be.p(be.buf(w1)), 32, be.p(be.buf(wh)), be.p(be.buf(wv)),
                              be.p(be.buf(cx)), be.p(be.buf(cy)), be.p(be.buf(bk1)), be.p(be.buf(bkhv)), be.p(bidx))
    assert rc == 0
    got = pk.unpack_tiles(ctx.numpy(), B * S_cap, H * 64).reshape(B, S_cap, H, 64).transpose(0, 2, 1, 3)[:, :, :S]
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-2)
    assert np.abs(got - ref).mean() < 2e-3
    # the default 8-wave form (one query tile per wave) against the form with two tiles per wave / 4 waves: the same bits
    first = np.array(ctx.numpy(), copy=True)
    try:
        be.lib.mgk_set_attention_qt(2)
        ctx1 = be.zeros((B * S_cap * H * 64,), np.uint16)
        assert be.lib.mgk_attention(be.stream, 0, be.p(Q), be.p(K), be.p(V), be.p(ctx1), B, H, S, S, S_cap, S_cap,
                                    be.p(be.buf(mask)), be.p(be.buf(w1)), 32, be.p(be.buf(wh)), be.p(be.buf(wv)),
                                    be.p(be.buf(cx)), be.p(be.buf(cy)), be.p(be.buf(bk1)), be.p(be.buf(bkhv)), be.p(bidx)) == 0
        one = pk.unpack_tiles(ctx1.numpy(), B * S_cap, H * 64).reshape(B, S_cap, H, 64).transpose(0, 2, 1, 3)[:, :, :S]
    finally:
        be.lib.mgk_set_attention_qt(1)
    two = pk.unpack_tiles(first, B * S_cap, H * 64).reshape(B, S_cap, H, 64).transpose(0, 2, 1, 3)[:, :, :S]
    assert np.array_equal(one.view(np.uint32), two.view(np.uint32))


@pytest.mark.parametrize("be_name", BACKENDS)
def test_attention_decoder_self_and_cross(be_name):
    import torch
    from oracle.udop_oracle import relative_position_bucket as rpb
    be = get_backend(be_name)
    B, H, T, T_cap, Sk, Sk_cap = 2, 2, 40, 64, 100, 128
    q, k, v = [pk.bf16_round(rnd((B, H, T_cap, 64), 40 + i, 0.5)) for i in range(3)]
    wd = rnd((32, H), 45)
    dist = torch.arange(0, 64)
    tab = wd[rpb(-dist, False, 32, 128).numpy()].astype(np.float32)       # [64][H] indexed by i-j
    dmask = np.ones((B, T_cap), np.uint8)
    dmask[1, 30:] = 0
    ref = np.zeros((B, H, T, 64), np.float32)
    ii, jj = np.arange(T)[:, None], np.arange(T)[None, :]
    for b in range(B):
        for h in range(H):
            sc = q[b, h, :T] @ k[b, h, :T].T + tab[np.clip(ii - jj, 0, 63), h]
            sc = np.where((jj <= ii) & (dmask[b, None, :T] != 0), sc, -1e30)
            ref[b, h] = softmax_ref(sc) @ v[b, h, :T]
    Q, K, V = be.buf(pack_heads_rows(q)), be.buf(pack_heads_rows(k)), be.buf(pack_heads_t(v))
    ctx = be.zeros((B * T_cap * H * 64,), np.uint16)
    assert be.lib.mgk_attention(be.stream, 1, be.p(Q), be.p(K), be.p(V), be.p(ctx), B, H, T, T, T_cap, T_cap,
                                be.p(be.buf(dmask)), be.p(be.buf(tab)), 64, None, None, None, None, None, None, None) == 0
    got = pk.unpack_tiles(ctx.numpy(), B * T_cap, H * 64).reshape(B, T_cap, H, 64).transpose(0, 2, 1, 3)[:, :, :T]
    valid = np.ones((B, T), bool)
    valid[1, 30:] = True   # rows whose every key is masked still see key 0..i with dmask -> compare all defined rows
    np.testing.assert_allclose(got[0], ref[0], rtol=0, atol=2e-2)
    np.testing.assert_allclose(got[1, :, :30], ref[1, :, :30], rtol=0, atol=2e-2)
    # cross
    kx, vx = [pk.bf16_round(rnd((B, H, Sk_cap, 64), 50 + i, 0.5)) for i in range(2)]
    xm = np.ones((B, Sk_cap), np.uint8)
    xm[0, 7:20] = 0
    xm[:, Sk:] = 0
    refx = np.zeros((B, H, T, 64), np.float32)
    for b in range(B):
        for h in range(H):
            sc = q[b, h, :T] @ kx[b, h, :Sk].T
            sc = np.where(xm[b, None, :Sk] != 0, sc, -1e30)
            refx[b, h] = softmax_ref(sc) @ vx[b, h, :Sk]
    KX, VX = be.buf(pack_heads_rows(kx)), be.buf(pack_heads_t(vx))
    ctx2 = be.zeros((B * T_cap * H * 64,), np.uint16)
    assert be.lib.mgk_attention(be.stream, 2, be.p(Q), be.p(KX), be.p(VX), be.p(ctx2), B, H, T, Sk, T_cap, Sk_cap,
                                be.p(be.buf(xm)), None, 0, None, None, None, None, None, None, None) == 0
    got2 = pk.unpack_tiles(ctx2.numpy(), B * T_cap, H * 64).reshape(B, T_cap, H, 64).transpose(0, 2, 1, 3)[:, :, :T]
    np.testing.assert_allclose(got2, refx, rtol=0, atol=2e-2)


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("B,H,S", [(1, 2, 256), (8, 3, 512)])
def test_attention_without_bias_over_full_rows(be_name, B, H, S):
    """Bias-free attention over full rows of keys with no mask (the ChemicalOCR vision tower: every frame's patches attend all of them) takes the
    encoder's second-form kernel without its bias path (attention_enc_kernel<.., PLAIN>): against numpy, and against the first-form kernel the
    same call takes when a key mask is passed (all ones)."""
    if be_name == "emu" and B > 1:
        B, S = 2, 256
    be = get_backend(be_name)
    q, k, v = [pk.bf16_round(rnd((B, H, S, 64), 70 + i, 0.5)) for i in range(3)]
    ref = np.zeros((B, H, S, 64), np.float32)
    for b in range(B):
        for h in range(H):
            ref[b, h] = softmax_ref(q[b, h] @ k[b, h].T) @ v[b, h]
    Q, K, V = be.buf(pack_heads_rows(q)), be.buf(pack_heads_rows(k)), be.buf(pack_heads_t(v))
    out = []
    for mask in (None, np.ones((B, S), np.uint8)):
        ctx = be.zeros((B * S * H * 64,), np.uint16)
        assert be.lib.mgk_attention(be.stream, 2, be.p(Q), be.p(K), be.p(V), be.p(ctx), B, H, S, S, S, S,
                                    be.p(be.buf(mask)) if mask is not None else None, None, 0, None, None, None, None, None, None, None) == 0
        out.append(pk.unpack_tiles(ctx.numpy(), B * S, H * 64).reshape(B, S, H, 64).transpose(0, 2, 1, 3).copy())
    np.testing.assert_allclose(out[0], ref, rtol=0, atol=2e-2)
    np.testing.assert_allclose(out[1], ref, rtol=0, atol=2e-2)
    assert np.abs(out[0] - out[1]).max() < 1e-2


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("group", [1, 5])
def test_attention_step(be_name, group):
    be = get_backend(be_name)
    owners, H, cap = 3, 2, 96
    rows = owners * group
    q = pk.bf16_round(rnd((rows, H, 64), 60, 0.5))
    kc, vc = [pk.bf16_round(rnd((owners, H, cap, 64), 61 + i, 0.5)) for i in range(2)]
    lens = np.array([96, 37, 1], np.int32)
    bias = rnd((cap, H), 63) if group == 1 else None
    t = 40
    ref = np.zeros((rows, H, 64), np.float32)
    for r in range(rows):
        o = r // group
        n = (t + 1) if group == 1 else lens[o]
        for h in range(H):
            sc = kc[o, h, :n] @ q[r, h]
            if bias is not None:
                sc = sc + bias[t - np.arange(n), h]
            p = softmax_ref(sc[None])[0]
            ref[r, h] = p @ vc[o, h, :n]
    ctx = be.zeros((((rows + 31) // 32 * 32) * H * 64,), np.uint16)
    rc = be.lib.mgk_attention_step(be.stream, be.p(be.buf(pk.bf16_bits(q))), be.p(be.buf(pk.bf16_bits(kc))),
                                   be.p(be.buf(pk.bf16_bits(vc))), be.p(ctx), rows, H, group, cap,
                                   None if group == 1 else be.p(be.buf(lens)), t + 1,
                                   be.p(be.buf(bias)) if bias is not None else None, None, t)
    assert rc == 0
    got = pk.unpack_tiles(ctx.numpy(), rows, H * 64).reshape(rows, H, 64)
    np.testing.assert_allclose(got, ref, rtol=1 / 128, atol=2e-3)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_greedy_select(be_name):
    be = get_backend(be_name)
    rows, V, ldl, max_len = 5, 1000, 1024, 8
    lg = rnd((rows, ldl), 70)
    lg[0, 17] = lg[0, 900] = 9.0          # tie -> lowest index
    lg[1, 1] = 12.0                        # EOS wins
    lg[2, 1] = 12.0                        # EOS wins but row already finished -> pad
    lg[3, 1] = 12.0                        # EOS suppressed by min_length in the second call
    unf = np.array([1, 1, 0, 1, 1], np.int32)
    nxt, out = be.zeros((rows,), np.int64), be.zeros((rows, max_len), np.int64)
    ub, nu, t2 = be.buf(unf), be.zeros((1,), np.int32), be.zeros((rows, 2), np.float32)
    L = be.buf(lg)
    assert be.lib.mgk_greedy_select(be.stream, be.p(L), rows, V, ldl, 1, 0, 0, be.p(nxt), be.p(out), max_len, 3, be.p(ub),
                                    be.p(nu), be.p(t2)) == 0
    exp = np.array([17, 1, 0, 1, int(np.argmax(lg[4, :V]))])
    assert np.array_equal(nxt.numpy(), exp) and np.array_equal(out.numpy()[:, 3], exp)
    assert np.array_equal(ub.numpy(), [1, 0, 0, 0, 1]) and nu.numpy()[0] == 2
    srt = np.sort(lg[4, :V])
    np.testing.assert_allclose(t2.numpy()[4], [srt[-1], srt[-2]])
    ub2 = be.buf(unf)
    assert be.lib.mgk_greedy_select(be.stream, be.p(L), rows, V, ldl, 1, 0, 6, be.p(nxt), be.p(out), max_len, 3, be.p(ub2),
                                    be.p(nu), None) == 0
    l3 = lg[3, :V].copy(); l3[1] = -np.inf
    assert nxt.numpy()[3] == int(np.argmax(l3)) and ub2.numpy()[3] == 1


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("with_mask", [True, False])
def test_embed_assemble_matches_oracle(be_name, with_mask):
    import torch
    from markushgrapher_amd import synth
    from oracle.udop_oracle import Oracle
    from tests.conftest import load_golden
    be = get_backend(be_name)
    g = load_golden("g0_tiny.npz")
    shape = synth.SHAPES["tiny"]
    sd = synth.recipe_state_dict(shape, gain=1.0)
    o = Oracle(shape, sd)
    ids, bbox, am = g["input_ids"], g["bbox"], g["attention_mask"]
    B, L = ids.shape
    P, d, n = shape.num_patches, shape.d_model, shape.image_size // shape.patch_size
    S, S_cap = L + P, 64
    img = pk.bf16_round(rnd((B, P, d), 80))     # stand-in patch embeddings
    tok = o.w["shared.weight"][torch.from_numpy(ids)]
    emb, bbox64, mask = o.combine(torch.from_numpy(img), tok, torch.from_numpy(bbox), torch.from_numpy(am) if with_mask else None)
    cell, _ = o.cell_embed(bbox64)
    ref = (emb + cell).numpy()
    ref_mask = mask.numpy() if with_mask else np.ones((B, S), np.int64)
    hid, cx, cy = be.zeros((B, S_cap, d), np.float32), be.zeros((B, S_cap), np.float64), be.zeros((B, S_cap), np.float64)
    mk, xrow, xlen, err = be.zeros((B, S_cap), np.uint8), be.zeros((B, S_cap), np.int32), be.zeros((B,), np.int32), be.zeros((1,), np.int32)
    be.lib.mgk_embed_meta_bytes.restype = C.c_size_t
    meta = be.zeros((be.lib.mgk_embed_meta_bytes(B, S_cap),), np.uint8)
    rc = be.lib.mgk_embed_assemble(
        be.stream, be.p(meta), be.p(be.buf(ids)), be.p(be.buf(bbox)), be.p(be.buf(am.astype(np.uint8))) if with_mask else None,
        be.p(be.buf(img)), be.p(be.buf(pk.bf16_bits(sd["shared.weight"]))),
        be.p(be.buf(pk.bf16_bits(sd["encoder.cell_2d_embedding.x_position_embeddings.weight"]))),
        be.p(be.buf(pk.bf16_bits(sd["encoder.cell_2d_embedding.y_position_embeddings.weight"]))),
        B, L, P, d, n, shape.max_2d_position_embeddings, shape.vocab_size, S_cap,
        be.p(hid), be.p(cx), be.p(cy), be.p(mk), be.p(xrow), be.p(xlen), be.p(err))
    assert rc == 0 and err.numpy()[0] == 0
    np.testing.assert_allclose(hid.numpy()[:, :S], ref, rtol=0, atol=1e-6)
    assert np.all(hid.numpy()[:, S:] == 0)
    assert np.array_equal(mk.numpy()[:, :S], ref_mask.astype(np.uint8)) and np.all(mk.numpy()[:, S:] == 0)
    np.testing.assert_array_equal(cx.numpy()[:, :S], bbox64[:, :, [0, 2]].mean(-1).numpy())
    np.testing.assert_array_equal(cy.numpy()[:, :S], bbox64[:, :, [1, 3]].mean(-1).numpy())
    for b in range(B):
        att = np.flatnonzero(mk.numpy()[b])
        assert xlen.numpy()[b] == len(att)
        assert np.array_equal(xrow.numpy()[b, att], np.arange(len(att)))
        assert np.all(xrow.numpy()[b, mk.numpy()[b] == 0] == -1)


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("M,N,K,KS", [(32, 96, 256, 3), (20, 64, 128, 2), (70, 128, 512, 8)])
def test_gemm_splitk_add_norm_relu(be_name, M, N, K, KS):
    be = get_backend(be_name)
    x, w = rnd((M, K), 90), rnd((N, K), 91, 0.3)
    ref = pk.bf16_round(x) @ pk.bf16_round(w).T
    X, W = be.buf(pk.pack_tiles(x)), be.buf(pk.pack_tiles(w))
    Mp = (M + 31) // 32 * 32
    P = be.zeros((KS, Mp, N), np.float32)
    assert be.lib.mgk_gemm_splitk(be.stream, be.p(X), be.p(W), be.p(P), M, N, K, N, C.c_size_t(Mp * N), KS) == 0
    np.testing.assert_allclose(P.numpy()[:, :M].sum(0), ref, rtol=1e-4, atol=1e-4)
    # relu_pack
    y = be.zeros((Mp * N,), np.uint16)
    assert be.lib.mgk_relu_pack(be.stream, be.p(P), KS, N, C.c_size_t(Mp * N), be.p(y), M, N) == 0
    np.testing.assert_allclose(pk.unpack_tiles(y.numpy(), M, N), np.maximum(ref, 0), rtol=1 / 128, atol=1e-3)
    # fused residual add + RMSNorm + pack (d = N)
    h0, g = rnd((M, N), 92), 1 + 0.2 * rnd((N,), 93)
    h = be.buf(h0)
    xp = be.zeros((Mp * N,), np.uint16)
    assert be.lib.mgk_add_norm_pack(be.stream, be.p(h), be.p(P), KS, N, C.c_size_t(Mp * N), be.p(be.buf(g)), be.p(xp), M, N,
                                    C.c_float(1e-6), C.c_float(1.0)) == 0
    hn = h0 + ref
    np.testing.assert_allclose(h.numpy(), hn, rtol=1e-4, atol=1e-4)
    xr = g * (hn / np.sqrt((hn ** 2).mean(-1, keepdims=True) + 1e-6))
    np.testing.assert_allclose(pk.unpack_tiles(xp.numpy(), M, N), xr, rtol=1 / 100, atol=2e-3)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_beam_reorder_physical_copy(be_name):
    """mg_beam_reorder = index_select(0, beam_idx) on every layer's K and V (stock cache_utils.py:100-104)."""
    be = get_backend(be_name)
    layers, rows, H, cap, used = 2, 10, 2, 16, 5
    src = np.random.RandomState(3).randint(0, 65535, (layers, 2, rows, H, cap, 64)).astype(np.uint16)
    idx = np.random.RandomState(4).randint(0, rows, (rows,)).astype(np.int32)
    dst = be.zeros(src.shape, np.uint16)
    rc = be.lib.mg_beam_reorder(be.stream, be.p(be.buf(src)), be.p(dst), be.p(be.buf(idx)), layers, rows, H, cap, used)
    assert rc == 0
    got = dst.numpy()
    assert np.array_equal(got[:, :, :, :, :used], src[:, :, idx][:, :, :, :, :used])
    assert np.all(got[:, :, :, :, used:] == 0)


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("M,N,K", [(32, 64, 128), (20, 128, 256), (70, 64, 4096), (32, 64, 4096), (20, 128, 4096)])
def test_gemm_resid_deferred_norm(be_name, M, N, K):
    """h += X W^T; x = bf16(h*gain*gscale) un-normalised; per-row partial sums of squares; and a consumer that applies
    the deferred rsqrt(mean(h^2)+eps) must reproduce RMSNorm(h)*gain followed by the projection."""
    be = get_backend(be_name)
    be.lib.mgk_gemm_resid.argtypes = [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_float, C.c_float]
    x, w = rnd((M, K), 100), rnd((N, K), 101, 0.1)
    h0, g = rnd((M, N), 102), 1 + 0.2 * rnd((N,), 103)
    ref_h = h0 + pk.bf16_round(x) @ pk.bf16_round(w).T
    Mp = (M + 31) // 32 * 32
    h = be.buf(h0)
    xp = be.zeros((Mp * N,), np.uint16)
    part = be.zeros((Mp, N // 8), np.float32)
    assert be.lib.mgk_gemm_resid(be.stream, be.p(be.buf(pk.pack_tiles(x))), be.p(be.buf(pk.pack_tiles(w))), be.p(h), be.p(be.buf(g)),
                                 0.5, be.p(xp), be.p(part), M, N, K, None, 0, 0.0, 0.0) == 0
    tol = 1e-4 * max(1.0, np.sqrt(K / 128))
    np.testing.assert_allclose(h.numpy(), ref_h, rtol=1e-4, atol=tol)
    np.testing.assert_allclose(pk.unpack_tiles(xp.numpy(), M, N), ref_h * g * 0.5, rtol=1 / 100, atol=2e-3)
    np.testing.assert_allclose(part.numpy()[:M].sum(1), (ref_h ** 2).sum(1), rtol=1e-4)
    # consumer with the deferred scale: second residual projection W2 on x (gscale folded back out by construction)
    w2 = rnd((64, N), 104, 0.1)
    h2 = be.zeros((M, 64), np.float32)
    xp2 = be.zeros((Mp * 64,), np.uint16)
    part2 = be.zeros((Mp, 8), np.float32)
    g2 = np.ones((64,), np.float32)
    assert be.lib.mgk_gemm_resid(be.stream, be.p(xp), be.p(be.buf(pk.pack_tiles(w2))), be.p(h2), be.p(be.buf(g2)), 1.0, be.p(xp2),
                                 be.p(part2), M, 64, N, be.p(part), N // 8, 1.0 / N, 1e-6) == 0
    xn = pk.bf16_round(ref_h * g * 0.5) / np.sqrt((ref_h ** 2).mean(-1, keepdims=True) + 1e-6)
    np.testing.assert_allclose(h2.numpy(), xn @ pk.bf16_round(w2).T, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("M,N2,d,inner", [(32, 128, 64, 128), (45, 2048, 64, 64), (70, 256, 128, 64)])
def test_gemm_pair_product_weights(be_name, M, N2, d, inner):
    """Pair projection: the residual projection h += ctx Wr^T and, in the same launch, relu(Wn G (h + ctx Wr^T))
    computed as [Wn G | Wn G Wr] [bf16(h) ; ctx] from the product weight.  N2 = 2048 takes the whole-tile workgroup form,
    the others the half-tile form; M = 45 / 70 cover two and three row tiles."""
    be = get_backend(be_name)
    be.lib.mgk_gemm_pair.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p] * 7 + [C.c_int, C.c_int]
    Wn, Wr = rnd((N2, d), 200, 0.3), rnd((d, inner), 201, 0.3)
    gain = 1 + 0.2 * rnd((d,), 202)
    h0, ctx = rnd((M, d), 203), rnd((M, inner), 204)
    Wn_b, Wr_b, ctx_b, hb = pk.bf16_round(Wn), pk.bf16_round(Wr), pk.bf16_round(ctx), pk.bf16_round(h0)
    Mp = (M + 31) // 32 * 32
    xwin = np.zeros((Mp, d + inner), np.float32)
    xwin[:M, :d], xwin[:M, d:] = hb, ctx_b
    W2 = be.zeros((N2 * (d + inner),), np.uint16)
    scratch = be.zeros((N2 * (2 * d + inner) + d * inner,), np.float32)
    h = be.buf(h0)
    hb_out = be.zeros((Mp * d,), np.uint16)
    part = be.zeros((Mp, d // 8), np.float32)
    out2 = be.zeros((Mp * N2,), np.uint16)
    rc = be.lib.mgk_gemm_pair(be.stream, be.p(be.buf(pk.pack_tiles(Wn))), be.p(be.buf(pk.pack_tiles(Wr))), be.p(be.buf(gain)), N2, d, inner,
                              be.p(W2), be.p(scratch), be.p(be.buf(pk.pack_tiles(xwin))), be.p(h), be.p(hb_out), be.p(part), be.p(out2), M, 1)
    assert rc == 0
    # product weight: fp32 product of the bf16 factors, rounded to bf16 once
    w2_ref = np.concatenate([Wn_b * gain, (Wn_b * gain) @ Wr_b], axis=1)
    np.testing.assert_allclose(pk.unpack_tiles(W2.numpy(), N2, d + inner), w2_ref, rtol=1 / 200, atol=1e-4)
    # residual projection, its bf16 copy and the partial sums of squares
    h_ref = h0 + ctx_b @ Wr_b.T
    np.testing.assert_allclose(h.numpy(), h_ref, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(pk.unpack_tiles(hb_out.numpy(), M, d), h_ref, rtol=1 / 200, atol=1e-3)
    np.testing.assert_allclose(part.numpy()[:M].sum(1), (h_ref ** 2).sum(1), rtol=1e-4)
    # second projection == the sequential form relu(Wn G h_new) up to the bf16 roundings of h, ctx and the product weight
    seq = np.maximum((h0 + ctx @ Wr.T) @ (Wn * gain).T, 0)
    got = pk.unpack_tiles(out2.numpy(), M, N2)
    scale = np.abs(seq).max()
    assert np.abs(got - seq).max() < 0.03 * scale, np.abs(got - seq).max() / scale
    # and exactly (to bf16 output rounding) the product form on the rounded operands
    exact = np.maximum(xwin[:M] @ pk.bf16_round(w2_ref).T, 0)
    np.testing.assert_allclose(got, exact, rtol=1 / 100, atol=2e-3 * scale)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_attention_encoder_skips_padded_stages(be_name):
    """Encoder attention as mg_encode runs it: key stages (64 keys) and query blocks (128 queries) without an attended
    position are skipped.  Attended query rows must match the masked reference; rows of a skipped query block are zero."""
    import torch
    from oracle.udop_oracle import relative_position_bucket as rpb
    be = get_backend(be_name)
    B, H, S, S_cap = 2, 2, 300, 320
    q, k, v = [pk.bf16_round(rnd((B, H, S_cap, 64), 120 + i, 0.5)) for i in range(3)]
    rs = np.random.RandomState(15)
    cx, cy = np.round(rs.rand(B, S_cap) * 64) / 64.0, rs.rand(B, S_cap)
    mask = np.ones((B, S_cap), np.uint8)
    mask[0, 40:200] = 0          # stages 1 and 2 (keys 64..191) fully padded, stages 0 and 3 partly; query block 0 partly
    mask[1, 128:256] = 0         # stages 2, 3 and the whole query block 1 padded
    mask[:, S:] = 0
    w1, wh, wv = [rnd((32, H), 130 + i) for i in range(3)]
    pos = np.arange(S)
    b1 = rpb(torch.from_numpy(pos[None, :] - pos[:, None]), True, 32, 128).numpy()
    ref = np.zeros((B, H, S, 64), np.float32)
    for b in range(B):
        dx = ((torch.from_numpy(cx[b, None, :S] - cx[b, :S, None]).float() * 100).to(torch.long))
        dy = ((torch.from_numpy(cy[b, None, :S] - cy[b, :S, None]).float() * 100).to(torch.long))
        bh, bv = rpb(dx, True, 32, 100).numpy(), rpb(dy, True, 32, 100).numpy()
        for h in range(H):
            sc = q[b, h, :S] @ k[b, h, :S].T + w1[b1, h] + wh[bh, h] + wv[bv, h]
            sc = np.where(mask[b, None, :S] != 0, sc, -1e30)
            ref[b, h] = softmax_ref(sc) @ v[b, h, :S]
    Q, K, V = be.buf(pack_heads_rows(q)), be.buf(pack_heads_rows(k)), be.buf(pack_heads_t(v))
    ctx = be.buf(np.full((B * S_cap * H * 64,), 0x7FC0, np.uint16))       # NaN pattern: skipped blocks must clear it
    bk1 = rpb(torch.arange(-128, 129), True, 32, 128).numpy().astype(np.int32)
    bkhv = rpb(torch.arange(-100, 101), True, 32, 100).numpy().astype(np.int32)
    bidx = be.zeros((B * S_cap * S_cap,), np.uint16)
    kst = be.zeros((B * (1 + S_cap // 64),), np.int32)
    qbv = be.zeros((B * ((S_cap + 127) // 128),), np.uint8)
    be.lib.mgk_attention_enc_skip.argtypes = [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p] * 11
    rc = be.lib.mgk_attention_enc_skip(be.stream, be.p(Q), be.p(K), be.p(V), be.p(ctx), B, H, S, S_cap, be.p(be.buf(mask)),
                                       be.p(be.buf(w1)), be.p(be.buf(wh)), be.p(be.buf(wv)), be.p(be.buf(cx)), be.p(be.buf(cy)),
                                       be.p(be.buf(bk1)), be.p(be.buf(bkhv)), be.p(bidx), be.p(kst), be.p(qbv))
    assert rc == 0
    ks = kst.numpy().reshape(B, 1 + S_cap // 64)
    assert ks[0, 0] == 3 and ks[0, 1:4].tolist() == [0, 3, 4] and ks[1, 0] == 3 and ks[1, 1:4].tolist() == [0, 1, 4]
    assert qbv.numpy().reshape(B, 3).tolist() == [[1, 1, 1], [1, 0, 1]]
    got = pk.unpack_tiles(ctx.numpy(), B * S_cap, H * 64).reshape(B, S_cap, H, 64).transpose(0, 2, 1, 3)
    for b in range(B):
        rows = np.nonzero(mask[b, :S])[0]
        np.testing.assert_allclose(got[b][:, rows], ref[b][:, rows], rtol=0, atol=2e-2)
    assert np.all(got[1][:, 128:256] == 0)            # the skipped query block was cleared
    assert np.all(np.isfinite(got[:, :, :S]))


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("M,N,K", [(70, 64, 256), (128, 96 + 32, 4096), (150, 64, 256)])
def test_residual_projection_feature_width_forms_are_bit_identical(be_name, M, N, K):
    """Residual projections with several row tiles: 16 output features per workgroup (the default: half the activation bytes through
    L2) against 8 (the one-tile forms' width).  Every output element is the same chain of MFMAs and the partial sums of squares keep
    their 8-feature groups: h, the packed bf16(h * gain) and the partial sums must be the SAME BITS; K = 4096 takes the 16-wave form."""
    if be_name == "emu" and K > 1024:
        M = 64
    be = get_backend(be_name)
    be.lib.mgk_gemm_resid.argtypes = [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_float, C.c_float]
    x, w = rnd((M, K), 310), rnd((N, K), 311, 0.1)
    h0, g = rnd((M, N), 312), 1 + 0.2 * rnd((N,), 313)
    Mp = (M + 31) // 32 * 32
    Xp, Wp, G = be.buf(pk.pack_tiles(x)), be.buf(pk.pack_tiles(w)), be.buf(g)
    res = {}
    try:
        for f16 in (0, 1):
            assert be.lib.mgk_set_resid_f16(f16) == 0
            h = be.buf(h0)
            xp = be.zeros((Mp * N,), np.uint16)
            part = be.zeros((Mp, N // 8), np.float32)
            assert be.lib.mgk_gemm_resid(be.stream, be.p(Xp), be.p(Wp), be.p(h), be.p(G), 0.5, be.p(xp), be.p(part), M, N, K, None, 0, 0.0, 0.0) == 0
            res[f16] = [np.array(a.numpy(), copy=True) for a in (h, xp, part)]
    finally:
        be.lib.mgk_set_resid_f16(1)
    np.testing.assert_allclose(res[1][0], h0 + pk.bf16_round(x) @ pk.bf16_round(w).T, rtol=1e-4, atol=5e-4)
    np.testing.assert_allclose(res[1][2][:M].sum(1), (res[1][0] ** 2).sum(1), rtol=1e-4)
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_decode_projections_row_tile_split_modes_are_bit_identical(be_name):
    """More than one 32-row tile of live sequences (beam search, the OCR stage at large batch): the decode projections either walk
    their row tiles in one workgroup (mode 0) or run one tile per workgroup (mode 1: grid.y and shifted argument views, the default
    for small weights).  Both must give the SAME BITS (same arithmetic per row, same summation order)."""
    be = get_backend(be_name)
    be.lib.mgk_gemm_resid.argtypes = [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_float, C.c_float]
    M, N, K = 150, 64, 256                      # 5 row tiles, the last one partial
    x, w = rnd((M, K), 300), rnd((N, K), 301, 0.1)
    h0, g = rnd((M, N), 302), 1 + 0.2 * rnd((N,), 303)
    Mp = (M + 31) // 32 * 32
    Xp, Wp, G = be.buf(pk.pack_tiles(x)), be.buf(pk.pack_tiles(w)), be.buf(g)
    w3 = rnd((96, K), 304, 0.1)
    W3 = be.buf(pk.pack_tiles(w3))
    results = {}
    try:
        for mode in (0, 1):
            assert be.lib.mgk_set_rows_split(mode) == 0
            h = be.buf(h0)
            xp = be.zeros((Mp * N,), np.uint16)
            part = be.zeros((Mp, N // 8), np.float32)
            assert be.lib.mgk_gemm_resid(be.stream, be.p(Xp), be.p(Wp), be.p(h), be.p(G), 0.5, be.p(xp), be.p(part), M, N, K, None, 0, 0.0, 0.0) == 0
            out_f = be.zeros((M, 96), np.float32)
            out_pk = be.zeros((Mp * 96,), np.uint16)
            assert be.lib.mgk_gemm(be.stream, 1, 0, be.p(Xp), be.p(W3), M, 96, K, be.p(out_f), 96, None, None) == 0      # fp32 store
            assert be.lib.mgk_gemm(be.stream, 1, 2, be.p(Xp), be.p(W3), M, 96, K, None, 96, None, be.p(out_pk)) == 0    # packed relu
            results[mode] = [np.array(a.numpy(), copy=True) for a in (h, xp, part, out_f, out_pk)]
    finally:
        be.lib.mgk_set_rows_split(-1)
    ref_h = h0 + pk.bf16_round(x) @ pk.bf16_round(w).T
    np.testing.assert_allclose(results[0][0], ref_h, rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(results[0][3], pk.bf16_round(x) @ pk.bf16_round(w3).T, rtol=1e-4, atol=2e-4)
    for mode in (1,):
        for a, b in zip(results[0], results[mode]):
            assert np.array_equal(a, b), mode


def _tile_f32(h):
    """[M][N] fp32 -> the encoder's tiled residual layout [M/32][N/4][32][4] (mg_device.h ht_off), flattened."""
    M, N = h.shape
    return np.ascontiguousarray(h.reshape(M // 32, 32, N // 4, 4).transpose(0, 2, 1, 3)).reshape(-1)


def _untile_f32(t, M, N):
    return np.ascontiguousarray(np.asarray(t).reshape(M // 32, N // 4, 32, 4).transpose(0, 2, 1, 3)).reshape(M, N)


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("M,d,K,N2", [(320, 128, 64, 256), (320, 2048, 64, 256), (640, 1536, 128, 256)])
def test_gemm_encoder_deferred_norm(be_name, M, d, K, N2):
    """Encoder deferred RMSNorm (EPI_RESID_NORM + a row-scaled consumer), large-M tile kernels: h (tiled fp32) += X W^T,
    x = bf16(h * gain) un-normalised, part[m][d/64] partial sums of h^2; then relu(W2 x) * rsqrt(mean h^2 + eps) per row must
    equal relu(W2 (RMSNorm(h) * gain)).  d = 2048 / 1536 give 32 / 24 partial sums per row: more than one group of 16 in
    row_scales_tiles (they were silently truncated to the first 16)."""
    if be_name == "emu" and d > 128:
        M = 320                                         # the emulator checks the index math of the wide case on one block row
    be = get_backend(be_name)
    be.lib.mgk_gemm_norm.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p] * 4 + \
                                    [C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float]
    x, w = rnd((M, K), 300), rnd((d, K), 301, 0.2)
    h0, g = rnd((M, d), 302), 1 + 0.2 * rnd((d,), 303)
    h0[:, d // 2:] *= 3.0                                # the late partial sums carry most of the energy: truncation would show
    ref_h = h0 + pk.bf16_round(x) @ pk.bf16_round(w).T
    np4 = (d // 64 + 3) // 4 * 4
    h = be.buf(_tile_f32(h0))
    xo = be.zeros((M * d,), np.uint16)
    part = be.zeros((M, np4), np.float32)
    assert be.lib.mgk_gemm_norm(be.stream, 5, be.p(be.buf(pk.pack_tiles(x))), be.p(be.buf(pk.pack_tiles(w))), M, d, K, be.p(h),
                                be.p(be.buf(g)), be.p(xo), be.p(part), np4, None, 0, 0.0, 0.0) == 0
    np.testing.assert_allclose(_untile_f32(h.numpy(), M, d), ref_h, rtol=1e-4, atol=1e-4)
    xg = pk.unpack_tiles(xo.numpy(), M, d)
    np.testing.assert_allclose(xg, ref_h * g, rtol=1 / 128, atol=2e-3)
    np.testing.assert_allclose(part.numpy().sum(1), (ref_h ** 2).sum(1), rtol=1e-4)
    w2 = rnd((N2, d), 304, 0.1)
    y = be.zeros((M * N2,), np.uint16)
    assert be.lib.mgk_gemm_norm(be.stream, 2, be.p(xo), be.p(be.buf(pk.pack_tiles(w2))), M, N2, d, None, None, be.p(y), None, 0,
                                be.p(part), np4, 1.0 / d, 1e-6) == 0
    r = 1.0 / np.sqrt((ref_h ** 2).mean(-1, keepdims=True) + 1e-6)
    want = np.maximum((xg @ pk.bf16_round(w2).T) * r, 0)
    np.testing.assert_allclose(pk.unpack_tiles(y.numpy(), M, N2), want, rtol=1 / 100, atol=2e-3 * np.abs(want).max())


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("variant", [3, 4, 5])
@pytest.mark.parametrize("M,N,K", [(640, 256, 64), (1280, 512, 128)])
def test_gemm_row_tile_list(be_name, M, N, K, variant):
    """The encoder's row-tile list (GemmArgs::row_tiles): only 32-row tiles with an attended position are computed.  Live tiles must
    equal the full GEMM bit for bit (same kernel, same K order), dead tiles must be left untouched - for the fp32 store, the packed
    relu output with deferred row scales, and the tiled residual + packed + partial-sum epilogue."""
    be = get_backend(be_name)
    be.lib.mgk_gemm_set_variant(variant)          # (put back by the autouse fixture _default_gemm_variant)
    be.lib.mgk_gemm_row_tiles.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p] * 4 + \
                                         [C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    be.lib.mgk_gemm_norm.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p] * 4 + \
                                    [C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float]
    x, w = rnd((M, K), 400), rnd((N, K), 401, 0.2)
    nt = M // 32
    live = np.ones(nt, bool)
    live[[1, 2, 5, nt - 1]] = False                       # dead tiles inside a block, across a block boundary (tiles 9/10) and at the end
    live[9:12] = False
    mask = np.zeros(M, np.uint8)
    for t in np.nonzero(live)[0]:
        mask[32 * t + (7 * t) % 32] = 1                   # one attended position is enough to make a tile live
    rowlive = np.repeat(live, 32)
    X, W = be.buf(pk.pack_tiles(x)), be.buf(pk.pack_tiles(w))
    mk, scratch = be.buf(mask), be.zeros((nt + 1,), np.int32)
    ref = pk.bf16_round(x) @ pk.bf16_round(w).T
    # fp32 store
    out = be.buf(np.full((M, N), -7.0, np.float32))
    assert be.lib.mgk_gemm_row_tiles(be.stream, 0, be.p(X), be.p(W), M, N, K, be.p(out), None, None, None, 0, None, 0, 0.0, 0.0, be.p(mk), be.p(scratch)) == 0
    full = be.zeros((M, N), np.float32)
    assert be.lib.mgk_gemm(be.stream, 0, 0, be.p(X), be.p(W), M, N, K, be.p(full), N, None, None) == 0
    o, f = out.numpy(), full.numpy()
    assert np.array_equal(o[rowlive], f[rowlive]) and np.all(o[~rowlive] == -7.0)
    np.testing.assert_allclose(f, ref, rtol=1e-4, atol=1e-4)
    sc = scratch.numpy()
    assert sc[0] == live.sum() and np.array_equal(sc[1:1 + sc[0]], np.nonzero(live)[0])
    # tiled residual + packed x + partial sums (epi 5), then a row-scaled relu consumer (epi 2) on the same list
    d = N
    h0, g = rnd((M, d), 402), 1 + 0.2 * rnd((d,), 403)
    np4 = (d // 64 + 3) // 4 * 4
    res = {}
    for name, use_list in (("list", True), ("full", False)):
        h = be.buf(_tile_f32(h0))
        xo = be.buf(np.full((M * d,), 0x7fc0, np.uint16))           # bf16 NaN pattern: a dead tile's rows must not reach a live result
        part = be.zeros((M, np4), np.float32)
        y = be.buf(np.full((M * N,), 0x1234, np.uint16))
        if use_list:
            assert be.lib.mgk_gemm_row_tiles(be.stream, 5, be.p(X), be.p(W), M, d, K, be.p(h), be.p(be.buf(g)), be.p(xo), be.p(part), np4, None, 0,
                                             0.0, 0.0, be.p(mk), be.p(scratch)) == 0
            w2 = be.buf(pk.pack_tiles(rnd((N, d), 404, 0.1)))
            assert be.lib.mgk_gemm_row_tiles(be.stream, 2, be.p(xo), be.p(w2), M, N, d, None, None, be.p(y), None, 0, be.p(part), np4, 1.0 / d,
                                             1e-6, be.p(mk), be.p(scratch)) == 0
        else:
            assert be.lib.mgk_gemm_norm(be.stream, 5, be.p(X), be.p(W), M, d, K, be.p(h), be.p(be.buf(g)), be.p(xo), be.p(part), np4, None, 0, 0.0, 0.0) == 0
            w2 = be.buf(pk.pack_tiles(rnd((N, d), 404, 0.1)))
            assert be.lib.mgk_gemm_norm(be.stream, 2, be.p(xo), be.p(w2), M, N, d, None, None, be.p(y), None, 0, be.p(part), np4, 1.0 / d, 1e-6) == 0
        res[name] = (_untile_f32(h.numpy(), M, d).copy(), pk.unpack_tiles(xo.numpy(), M, d).view(np.uint32).copy(), part.numpy().copy(),
                     pk.unpack_tiles(y.numpy(), M, N).view(np.uint32).copy())
    for a_, b_ in zip(res["list"], res["full"]):
        assert np.array_equal(a_[rowlive], b_[rowlive])
    assert np.array_equal(res["list"][0][~rowlive], h0[~rowlive])                   # dead rows of the residual stream untouched
    assert np.all(res["list"][2][~rowlive] == 0)


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("parts", [2, 3])
def test_gemm_pp_row_tiles_over_several_launches(be_name, parts):
    """Ping-pong kernel, a problem's row tiles in several launches (what keeps the FFN input projection of a 160-image call on it: more
    tiles per workgroup than its table holds): every output tile is the same arithmetic whichever launch computes it - same bits as
    one launch, with and without a row-tile list, fp32 store and tiled residual + packed + partial-sum epilogue."""
    be = get_backend(be_name)
    be.lib.mgk_gemm_set_variant(5)                # (put back by the autouse fixture _default_gemm_variant)
    be.lib.mgk_gemm_row_tiles.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p] * 4 + \
                                         [C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    M, N, K = 1280 + 96, 512, 128
    x, w = rnd((M, K), 410), rnd((N, K), 411, 0.2)
    nt = (M + 31) // 32
    live = np.ones(nt, bool)
    live[[0, 3, 4, 17, nt - 2]] = False
    mask = np.zeros(nt * 32, np.uint8)
    for t in np.nonzero(live)[0]:
        mask[32 * t + (5 * t) % 32] = 1
    mask = mask[:M].copy()
    X, W = be.buf(pk.pack_tiles(x)), be.buf(pk.pack_tiles(w))
    h0, g = rnd((nt * 32, N), 412), 1 + 0.2 * rnd((N,), 413)
    np4 = (N // 64 + 3) // 4 * 4
    res = {}
    try:
        for mode in (1, parts):                   # (1: only when the kernel's tile table is too small - one launch at this size)
            assert be.lib.mgk_set_pp_parts(mode) == 0
            full = be.zeros((M, N), np.float32)
            assert be.lib.mgk_gemm(be.stream, 0, 0, be.p(X), be.p(W), M, N, K, be.p(full), N, None, None) == 0
            mk, scratch = be.buf(mask), be.zeros((nt + 1,), np.int32)
            out = be.buf(np.full((M, N), -7.0, np.float32))
            assert be.lib.mgk_gemm_row_tiles(be.stream, 0, be.p(X), be.p(W), M, N, K, be.p(out), None, None, None, 0, None, 0, 0.0, 0.0, be.p(mk), be.p(scratch)) == 0
            h = be.buf(_tile_f32(h0))
            xo = be.buf(np.full((nt * 32 * N,), 0x7fc0, np.uint16))
            part = be.zeros((nt * 32, np4), np.float32)
            assert be.lib.mgk_gemm_row_tiles(be.stream, 5, be.p(X), be.p(W), M, N, K, be.p(h), be.p(be.buf(g)), be.p(xo), be.p(part), np4, None, 0,
                                             0.0, 0.0, be.p(mk), be.p(scratch)) == 0
            res[mode] = (full.numpy().copy(), out.numpy().copy(), h.numpy().copy(), xo.numpy().copy(), part.numpy().copy())
    finally:
        be.lib.mgk_set_pp_parts(0)               # (the default)
    np.testing.assert_allclose(res[1][0], pk.bf16_round(x) @ pk.bf16_round(w).T, rtol=1e-4, atol=1e-4)
    rowlive = np.repeat(live, 32)[:M]
    assert np.all(res[1][1][~rowlive] == -7.0) and np.array_equal(res[1][1][rowlive], res[1][0][rowlive])
    for a_, b_ in zip(res[1], res[parts]):
        assert np.array_equal(a_.view(np.uint8), b_.view(np.uint8))


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("M,N,K", [(150, 256, 4096), (64, 256, 1024), (96, 512, 512)])
def test_residual_projection_k_slab_form_is_bit_identical(be_name, M, N, K):
    """Residual projection over several row tiles, K-slab form (gemm_rows_resid_mt_kernel: the K chunks of the one-workgroup form's
    waves become workgroups, the last arrival of a feature tile adds the partial sums in the waves' order): h, bf16(h * gain) and the
    partial sums of squares must be the SAME BITS as the one-workgroup form's, with a deferred row scale on the input, a partial last
    row tile, and twice in a row on the same scratch (the tickets return to zero)."""
    if be_name == "emu" and K > 1024:
        M = 40
    be = get_backend(be_name)
    at = [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    be.lib.mgk_gemm_resid_mt.argtypes = at
    x, w = rnd((M, K), 410), rnd((N, K), 411, 0.1)
    h0, g = rnd((M, N), 412), 1 + 0.2 * rnd((N,), 413)
    Mp = (M + 31) // 32 * 32
    rsp = np.abs(rnd((Mp, 16), 414)) + 0.5
    Xp, Wp, G, RS = be.buf(pk.pack_tiles(x)), be.buf(pk.pack_tiles(w)), be.buf(g), be.buf(rsp)
    res = {}
    try:
        # 0: one-workgroup form; 1: K-slab form, merged by the last arrival; 2: again on the same scratch (the tickets are back at zero);
        # 3: K-slab form in two launches (partial sums, then the chip-wide merge launch)
        for mode in (0, 1, 2, 3):
            assert be.lib.mgk_set_rows_mt({0: 0, 1: 1, 2: 1, 3: 2}[mode]) == 0
            if mode < 2:
                kpart = be.zeros((16 * Mp * N,), np.float32)
                ticket = be.zeros((N // 32,), np.int32)
            h = be.buf(h0)
            xp = be.zeros((Mp * N,), np.uint16)
            part = be.zeros((Mp, N // 8), np.float32)
            assert be.lib.mgk_gemm_resid_mt(be.stream, be.p(Xp), be.p(Wp), be.p(h), be.p(G), 0.5, be.p(xp), be.p(part), M, N, K, be.p(RS), 16,
                                            1.0 / 16, 1e-6, 8, be.p(kpart), be.p(ticket)) == 0
            res[mode] = [np.array(a.numpy(), copy=True) for a in (h, xp, part)]
            assert not ticket.numpy().any()
    finally:
        be.lib.mgk_set_rows_mt(0)            # (the default: the K-slab form is measured slower than the one-workgroup forms, k_gemm.hip)
    r = 1.0 / np.sqrt(rsp.sum(1)[:M] / 16 + 1e-6)
    np.testing.assert_allclose(res[1][0], h0 + (pk.bf16_round(x) @ pk.bf16_round(w).T) * r[:, None], rtol=1e-4, atol=5e-4)
    for mode in (1, 2, 3):
        for a, b in zip(res[0], res[mode]):
            va, vb = (a.view(np.uint32), b.view(np.uint32)) if a.dtype == np.float32 else (a, b)
            if a.ndim == 2 and a.shape[0] == Mp:
                va, vb = va[:M], vb[:M]
            assert np.array_equal(va, vb), mode


@pytest.mark.parametrize("be_name", BACKENDS)
@pytest.mark.parametrize("d,H,nsplit", [(64, 2, 1), (256, 4, 2), (1024, 16, 1), (768, 12, 3)])
def test_absorbed_cross_attention(be_name, d, H, nsplit):
    """k_xattn.hip against the stock formulation: K = enc Wk^T, V = enc Wv^T per head, softmax(q K^T) V (modeling_udop.py:524-575), fp32 numpy on
    the bf16-rounded operands.  Lengths cover one key, a partial stage, whole stages and the capacity; rows read owners through kv_owner."""
    be = get_backend(be_name)
    if be_name == "emu" and d >= 768:
        cap, lens = 48, np.array([48, 19], np.int32)
    else:
        cap, lens = 160, np.array([160, 37, 1, 16, 97], np.int32)
    owners = len(lens)
    owner_of = np.array([(3 * r + 1) % owners for r in range(owners + 2)], np.int32)
    rows = len(owner_of)
    inner = H * 64
    q = pk.bf16_round(rnd((rows, H, 64), 80, 0.5))
    wkv = pk.bf16_round(rnd((2 * inner, d), 81, 1.0 / np.sqrt(d)))
    enc = pk.bf16_round(rnd((owners, cap, d), 82, 1.0))
    ref = np.zeros((rows, H, 64), np.float32)
    for r in range(rows):
        o, n = owner_of[r], lens[owner_of[r]]
        for h in range(H):
            K = enc[o, :n] @ wkv[h * 64:(h + 1) * 64].T
            V = enc[o, :n] @ wkv[inner + h * 64:inner + (h + 1) * 64].T
            ref[r, h] = softmax_ref((K @ q[r, h])[None])[0] @ V
    ctx = be.zeros((((rows + 31) // 32 * 32) * inner,), np.uint16)
    wk, wv = be.zeros((H * d * 64,), np.uint16), be.zeros((H * d * 64,), np.uint16)
    qx = be.zeros((rows * H * d,), np.uint16)
    part, ml = be.zeros((rows * nsplit * H * d,), np.float32), be.zeros((rows * nsplit * H * 2,), np.float32)
    # both forms of the stream kernel: one wave group on a ring of three stages (what the engine runs), two wave groups on a ring of four
    gots = []
    for nstg in (3, 4):
        rc = be.lib.mgk_xattn(be.stream, be.p(be.buf(pk.bf16_bits(q))), be.p(be.buf(wkv)), be.p(be.buf(pk.bf16_bits(enc))), be.p(be.buf(lens)),
                              be.p(be.buf(owner_of)), rows, H, d, cap, nsplit, nstg, be.p(wk), be.p(wv), be.p(qx), be.p(part), be.p(ml), be.p(ctx))
        assert rc == 0
        gots.append(pk.unpack_tiles(ctx.numpy(), rows, inner).reshape(rows, H, 64).copy())
    assert np.abs(gots[0] - gots[1]).max() < 2e-2          # (the groups split the keys differently: same values up to rounding)
    got = gots[0]
    # q' = q Wk_h and the normalised context are rounded to bf16 (2^-9 relative each) where the K / V form rounds K and V: the
    # error against the fp32 formulation is held against what the K / V form's own roundings (K, V, P, ctx in bf16) cost on the same inputs
    kvf = np.zeros_like(ref)
    for r in range(rows):
        o, n = owner_of[r], lens[owner_of[r]]
        for h in range(H):
            K = pk.bf16_round(enc[o, :n] @ wkv[h * 64:(h + 1) * 64].T)
            V = pk.bf16_round(enc[o, :n] @ wkv[inner + h * 64:inner + (h + 1) * 64].T)
            kvf[r, h] = pk.bf16_round(pk.bf16_round(softmax_ref((K @ q[r, h])[None])[0]) @ V)
    e_abs, e_kv = np.abs(got - ref), np.abs(kvf - ref)
    assert e_abs.max() <= 2.0 * e_kv.max() + 2e-3 and e_abs.mean() <= 2.0 * e_kv.mean() + 2e-4, (e_abs.max(), e_kv.max(), e_abs.mean(), e_kv.mean())
    np.testing.assert_allclose(got, ref, rtol=1 / 32, atol=2e-2)
    # transpose-detecting: the result must not match the reference of another head / row
    assert np.abs(got - np.roll(ref, 1, axis=1)).max() > 0.05 and np.abs(got - np.roll(ref, 1, axis=0)).max() > 0.05


@pytest.mark.parametrize("be_name", BACKENDS)
def test_enc_rows_compaction(be_name):
    be = get_backend(be_name)
    B, S, cap, d = 3, 64, 80, 128
    x = pk.bf16_round(rnd((B * S, d), 90))
    rmap = np.full((B * S,), -1, np.int32)
    rs = np.random.RandomState(5)
    exp = np.zeros((B, cap, d), np.float32)
    for b in range(B):
        keep = np.sort(rs.choice(S, size=20 + 7 * b, replace=False))
        for j, s in enumerate(keep):
            rmap[b * S + s] = 5 + j
            exp[b, 5 + j] = x[b * S + s]
    dst = be.zeros((B * cap * d,), np.uint16)
    assert be.lib.mgk_enc_rows(be.stream, be.p(be.buf(pk.pack_tiles(x))), be.p(be.buf(rmap)), be.p(dst), B, S, cap, d) == 0
    got = pk.bf16_to_f32(dst.numpy()).reshape(B, cap, d)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("be_name", BACKENDS)
def test_absorbed_cross_attention_ignores_what_lies_behind_the_last_key(be_name):
    """The stream reads whole stages of 16 keys: rows between an image's last key and the next multiple of 16 carry weight exactly 0 (the
    engine clears them, enc_pad_rows).  With finite junk there the result must be the SAME BITS as with zeros."""
    be = get_backend(be_name)
    d, H, cap, rows = 128, 2, 64, 3
    lens = np.array([37, 1, 50], np.int32)
    inner = H * 64
    q = pk.bf16_round(rnd((rows, H, 64), 90, 0.5))
    wkv = pk.bf16_round(rnd((2 * inner, d), 91, 1.0 / np.sqrt(d)))
    enc = pk.bf16_round(rnd((rows, cap, d), 92, 1.0))
    outs = []
    for junk in (0.0, 7.5):
        e = enc.copy()
        for r in range(rows):
            e[r, lens[r]:] = junk
        ctx = be.zeros((32 * inner,), np.uint16)
        wk, wv = be.zeros((H * d * 64,), np.uint16), be.zeros((H * d * 64,), np.uint16)
        qx, part, ml = be.zeros((rows * H * d,), np.uint16), be.zeros((rows * H * d,), np.uint16), be.zeros((rows * H * 2,), np.float32)
        assert be.lib.mgk_xattn(be.stream, be.p(be.buf(pk.bf16_bits(q))), be.p(be.buf(wkv)), be.p(be.buf(pk.bf16_bits(e))), be.p(be.buf(lens)), None,
                                rows, H, d, cap, 1, 4, be.p(wk), be.p(wv), be.p(qx), be.p(part), be.p(ml), be.p(ctx)) == 0
        outs.append(np.array(ctx.numpy(), copy=True))
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("be_name", BACKENDS)
def test_half_tile_projections_both_halves_form_is_bit_identical(be_name):
    """Half-tile decode projections with three or more row tiles of live rows: a workgroup takes BOTH 16-feature halves of its weight tile
    (rows_block16 FT = 2: every activation fragment feeds two MFMAs, half the activation re-reads through L2, half the workgroups) - the
    default - against one half per workgroup.  Same K partition over the waves, same reduction order: packed (plain / relu), fp32 and
    per-head outputs (q + cache append) must be the SAME BITS; values against the fp32 reference."""
    be = get_backend(be_name)
    M, K, N, H, T = (160, 512, 128, 2, 8) if be_name == "hip" else (100, 128, 64, 1, 4)
    inner = H * 64
    x, w, wq = rnd((M, K), 320), rnd((N, K), 321, 0.1), rnd((3 * inner, K), 322, 0.1)
    Mp = (M + 31) // 32 * 32
    X, W, WQ = be.buf(pk.pack_tiles(x)), be.buf(pk.pack_tiles(w)), be.buf(pk.pack_tiles(wq))
    res = {}
    try:
        assert be.lib.mgk_set_rows_split(0) == 0          # (small weights would otherwise take the one-row-tile-per-workgroup form)
        for ft2 in (0, 1):
            assert be.lib.mgk_set_rows_ft2(ft2) == 0
            out_f, out_pk, out_relu = be.zeros((M, N), np.float32), be.zeros((Mp * N,), np.uint16), be.zeros((Mp * N,), np.uint16)
            assert be.lib.mgk_gemm(be.stream, 1, 0, be.p(X), be.p(W), M, N, K, be.p(out_f), N, None, None) == 0
            assert be.lib.mgk_gemm(be.stream, 1, 3, be.p(X), be.p(W), M, N, K, None, N, None, be.p(out_pk)) == 0
            assert be.lib.mgk_gemm(be.stream, 1, 2, be.p(X), be.p(W), M, N, K, None, N, None, be.p(out_relu)) == 0
            q, kc, vc = be.zeros((M, H, 64), np.uint16), be.zeros((M, H, T, 64), np.uint16), be.zeros((M, H, T, 64), np.uint16)
            assert be.lib.mgk_gemm_heads(be.stream, 1, be.p(X), be.p(WQ), M, 3 * inner, K, be.p(q), be.p(kc), be.p(vc),
                                         HF_STEP_Q, HF_STEP_KV, HF_STEP_KV, H, M, T, None, 3) == 0
            res[ft2] = [np.array(a.numpy(), copy=True) for a in (out_f, out_pk, out_relu, q, kc, vc)]
    finally:
        be.lib.mgk_set_rows_ft2(-1)
        be.lib.mgk_set_rows_split(-1)
    ref = pk.bf16_round(x) @ pk.bf16_round(w).T
    np.testing.assert_allclose(res[1][0], ref, rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(pk.unpack_tiles(res[1][2], M, N), np.maximum(ref, 0), rtol=1 / 128, atol=1e-3)
    np.testing.assert_allclose(pk.bf16_to_f32(res[1][3]), (pk.bf16_round(x) @ pk.bf16_round(wq).T).reshape(M, 3, H, 64)[:, 0], rtol=1 / 128, atol=1e-3)
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
