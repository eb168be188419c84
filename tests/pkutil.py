"""numpy helpers for the tests: bf16 bit conversion and the packed fragment-tile format
(markushgrapher_amd/csrc/mg_device.h), restated independently of the HIP code."""
import numpy as np


def bf16_bits(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16).reshape(x.shape)


def bf16_to_f32(bits):
    bits = np.ascontiguousarray(bits, dtype=np.uint16)
    return (bits.astype(np.uint32) << 16).view(np.float32).reshape(bits.shape)


def bf16_round(x):
    return bf16_to_f32(bf16_bits(x))


def pack_tiles(x, rows_pad=None):
    """X[R][K] fp32 -> packed bf16 bits [R/32][K/16][2][32][8] flattened."""
    R, K = x.shape
    Rp = rows_pad if rows_pad is not None else (R + 31) // 32 * 32
    xb = np.zeros((Rp, K), np.uint16)
    xb[:R] = bf16_bits(x)
    t = xb.reshape(Rp // 32, 32, K // 16, 2, 8)          # [rt][row][kt][half][8]
    return np.ascontiguousarray(t.transpose(0, 2, 3, 1, 4)).reshape(-1)


def unpack_tiles(bits, R, K):
    """inverse of pack_tiles -> fp32 [R][K] (R multiple of 32 of the stored rows)."""
    Rp = bits.size // K
    t = np.asarray(bits, np.uint16).reshape(Rp // 32, K // 16, 2, 32, 8).transpose(0, 3, 1, 2, 4)
    return bf16_to_f32(np.ascontiguousarray(t).reshape(Rp, K))[:R]


def unpack_heads_rows(bits, B, H, S_cap):
    """HF_PK_ROWS [B][H][S_cap/32][4][2][32][8] -> fp32 [B][H][S_cap][64]"""
    t = np.asarray(bits, np.uint16).reshape(B, H, S_cap // 32, 4, 2, 32, 8).transpose(0, 1, 2, 5, 3, 4, 6)
    return bf16_to_f32(np.ascontiguousarray(t).reshape(B, H, S_cap, 64))


def unpack_heads_t(bits, B, H, S_cap):
    """HF_PK_T [B][H][2][S_cap/16][2][32][8] (rows = dim, k = token) -> fp32 [B][H][S_cap][64]"""
    t = np.asarray(bits, np.uint16).reshape(B, H, 2, S_cap // 16, 2, 32, 8)   # [b][h][dt][kt][half][dim32][8tok]
    t = t.transpose(0, 1, 3, 4, 6, 2, 5)                                       # [b][h][kt][half][8][dt][dim32]
    return bf16_to_f32(np.ascontiguousarray(t).reshape(B, H, S_cap, 64))
