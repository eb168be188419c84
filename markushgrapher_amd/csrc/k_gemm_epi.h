// Epilogues of the bf16 MFMA GEMM kernels (k_gemm.hip, k_gemm_pp.hip): row scales of the deferred RMSNorm, per-head stores, the
// tiled fp32 residual update, packed outputs.  Device code only; included by the GEMM translation units.
#pragma once
#include "mg_kernels.h"
#include <type_traits>

namespace mg {

// Deferred RMSNorm scales r(m) = rsqrt(sum_i part[m][i] * inv_d + eps) for rows [0, nrows) into LDS, computed by the
// whole workgroup: 8 threads per row, each summing nparts/8 partials with independent 16-byte loads (one L2 round
// trip, issued before the weight stream), combined by a fixed DPP tree.  Callers read `out` after their next barrier.
MG_DEV void block_row_scales(const RowScale& rs, int M, int nrows, float* out, int tid, int nthreads) {
    for (int base = 0; base < nrows; base += nthreads >> 3) {
        const int row = base + (tid >> 3), j = tid & 7;
        float s = 0.f;
        if (rs.part && row < nrows) {
            const int mr = row < M ? row : M - 1;
            const int per = rs.nparts >> 3;                               // floats per thread
            const float* p = rs.part + (size_t)mr * rs.nparts + j * per;
            if ((per & 3) == 0) {
                for (int i = 0; i < per; i += 4) { const float4 a = *(const float4*)(p + i); s += (a.x + a.y) + (a.z + a.w); }
            } else {
                for (int i = 0; i < per; ++i) s += p[i];
            }
        }
        s = sum8(s);
        if (j == 0 && row < nrows) out[row] = rs.part ? rsqrtf(s * rs.inv_d + rs.eps) : 1.0f;
    }
}

// Split form of block_row_scales for the common decode shape (all rows in one pass, <= 16 partials per thread): the
// partial sums are FETCHED first of all (rs_issue), the weight and activation streams are issued behind them, and the
// reduction (rs_finish) then waits only for these earliest loads - a wait on a later load would drain the whole in-order
// vector-memory queue, i.e. serialise the row scales behind the HBM round trip of the weights.
struct RsRegs { float4 v[4]; bool fast; };
MG_DEV void rs_issue(const RowScale& rs, int M, int nrows, int tid, int nthreads, RsRegs& r) {
    const int per = rs.nparts >> 3;
    r.fast = rs.part && nrows <= (nthreads >> 3) && per <= 16 && (per & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r.fast) {
        const int row = tid >> 3, j = tid & 7;
        if (row < nrows) {
            const int mr = row < M ? row : M - 1;
            const float* p = rs.part + (size_t)mr * rs.nparts + j * per;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * i < per) r.v[i] = *(const float4*)(p + 4 * i);
        }
    }
}
MG_DEV void rs_finish(const RowScale& rs, int M, int nrows, float* out, int tid, int nthreads, const RsRegs& r) {
    if (!r.fast) { block_row_scales(rs, M, nrows, out, tid, nthreads); return; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (r.v[i].x + r.v[i].y) + (r.v[i].z + r.v[i].w);
    s = sum8(s);
    const int row = tid >> 3;
    if ((tid & 7) == 0 && row < nrows) out[row] = rsqrtf(s * rs.inv_d + rs.eps);
}

// ---------------------------------------------------------------------------------------------------------
// epilogue helpers
// ---------------------------------------------------------------------------------------------------------
// One 16-byte chunk of a per-head projection.  `tok0..` semantics depend on the format:
//  token-major formats (PK_ROWS / NATURAL / STEP_*): chunk = token m, head dims [dim0, dim0+8)
//  HF_PK_T: chunk = head dim `dim0`, tokens [m, m+8)
MG_DEV void heads_store(const HeadsOut& ho, int ri, int h, int m, int dim0, const uint4& c) {
    const int fmt = ho.fmt[ri];
    uint16_t* base = ho.ptr[ri];
    if (fmt == HF_PK_ROWS) {
        const int b = m / ho.S_in, s = m - b * ho.S_in + ho.s_off;
        size_t off = (((size_t)b * ho.H + h) * (size_t)(ho.S_cap >> 5) + (size_t)(s >> 5)) * (4 * TILE_ELEMS) +
                     (size_t)(dim0 >> 4) * TILE_ELEMS + (size_t)(((dim0 >> 3) & 1) * 256 + (s & 31) * 8);
        st16(base + off, c);
    } else if (fmt == HF_PK_T) {
        const int b = m / ho.S_in, s = m - b * ho.S_in + ho.s_off;
        size_t off = ((((size_t)b * ho.H + h) * 2 + (size_t)(dim0 >> 5)) * (size_t)(ho.S_cap >> 4) + (size_t)(s >> 4)) *
                         TILE_ELEMS +
                     (size_t)(((s >> 3) & 1) * 256 + (dim0 & 31) * 8);
        st16(base + off, c);
    } else if (fmt == HF_NATURAL) {
        const int b = m / ho.S_in, s = m - b * ho.S_in;
        const int row = ho.row_map ? ho.row_map[m] : s;
        if (row >= 0) st16(base + (((size_t)b * ho.H + h) * (size_t)ho.S_cap + (size_t)row) * 64 + dim0, c);
    } else if (fmt == HF_STEP_Q) {
        st16(base + ((size_t)m * ho.H + h) * 64 + dim0, c);
    } else if (fmt == HF_STEP_KV) {
        const int row = ho.row_map ? ho.row_map[m] : m;
        const int pos = ho.pos_rows ? ho.pos_rows[m] : (ho.pos_dev ? *ho.pos_dev : ho.pos);
        st16_stream(base + (((size_t)row * ho.H + h) * (size_t)ho.S_cap + (size_t)pos) * 64 + dim0, c);
    }
}

// Epilogue of one 32x32 accumulator tile.
//  TOR (operands swapped, D = W·X^T): lane owns token m = m0 + lane%32, rows of D are output features n0 + i.
//  !TOR (D = X·W^T):                  lane owns feature n = n0 + lane%32, rows of D are tokens m0 + i.
// deferred RMSNorm scale of token row m from the partial sums of squares left by EPI_RESID_NORM (1 when rs.part is null)
MG_DEV float row_scale_of(const RowScale& rs, int m, int M) {
    if (!rs.part) return 1.0f;
    const int mr = m < M ? m : M - 1;
    const float* p = rs.part + (size_t)mr * rs.nparts;
    float s = 0.f;
    for (int i = 0; i < rs.nparts; i += 4) { const float4 v = *(const float4*)(p + i); s += (v.x + v.y) + (v.z + v.w); }
    return rsqrtf(s * rs.inv_d + rs.eps);
}

// the scales of NT consecutive 32-token row tiles for this lane's token (m0 + 32 i + lane%32): the partial-sum loads of a
// group of 16 partials are unconditional and issued together for all NT tiles (one L2 round trip at the start of the epilogue
// for d_model <= 1024; one more per further 1024 columns); nparts is a multiple of 4 (groups past nparts re-read the last one
// with weight 0)
template <int NT>
MG_DEV void row_scales_tiles(const RowScale& rs, const int (&mrow)[NT], int M, int lane, float (&out)[NT]) {
#pragma unroll
    for (int i = 0; i < NT; ++i) out[i] = 1.0f;
    if (!rs.part) return;
    float s[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) s[i] = 0.f;
    for (int g0 = 0; g0 < rs.nparts; g0 += 16) {
        float4 v[NT][4];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            int m = mrow[i] + (lane & 31);
            m = m < M ? m : M - 1;
            const float* p = rs.part + (size_t)m * rs.nparts;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[i][k] = *(const float4*)(p + (g0 + 4 * k < rs.nparts ? g0 + 4 * k : rs.nparts - 4));
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s[i] += (g0 + 4 * k < rs.nparts) ? (v[i][k].x + v[i][k].y) + (v[i][k].z + v[i][k].w) : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) out[i] = rsqrtf(s[i] * rs.inv_d + rs.eps);
}

// x Phi(x) with erf by Abramowitz & Stegun 7.1.26 (|error| < 1.5e-7 on erf, far below the bf16 rounding of the result): one reciprocal,
// one exp and seven fused multiply-adds instead of the library erff's branches - the epilogue of the Swin MLP's first projection
// evaluates it 16 times per lane and tile (stock ACT2FN["gelu"] = exact erf form)
MG_DEV float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = 1.0f / (1.0f + 0.3275911f * z);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float e = 1.0f - poly * fast_exp(-z * z);                   // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + (x < 0.f ? -e : e));
}

// APPLY_RS (tiled large-M kernels only; the decode-step kernels scale their sums themselves): multiply the rows of the
// packed / per-head outputs by the deferred RMSNorm scale a.rs of their token.
template <int EPI, bool TOR, bool APPLY_RS = false>
MG_DEV void tile_epilogue(const GemmArgs& a, const f32x16& acc_in, int m0, int n0, int lane, int qmask = 3, float rsl = 1.0f) {
    const int half = lane >> 5, l32 = lane & 31;
    f32x16 acc = acc_in;
    if constexpr (APPLY_RS && (EPI == EPI_PK || EPI == EPI_PK_RELU || EPI == EPI_PK_GELU || EPI == EPI_HEADS || EPI == EPI_PK_BIAS || EPI == EPI_PK_GELU_ERF)) {
        if (a.rs.part) {                                              // rsl: the scale of token m0 + lane%32 (both half-waves)
            if (TOR) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] *= rsl;
            } else {                                                   // rows of D are tokens m0 + acc_row(r, half)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] *= __shfl(rsl, acc_row(r, half));
            }
        }
    }
    if constexpr (EPI == EPI_F32_STORE || EPI == EPI_F32_RESID) {
        static_assert(!TOR, "fp32 epilogues use D = X·W^T");
        const int n = n0 + l32;
        if (n >= a.N) return;
        const float bv = (EPI == EPI_F32_STORE && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + acc_row(r, half);
            if (m < a.M) {
                float* p = a.out_f32 + (size_t)m * a.ldo + n;
                if (EPI == EPI_F32_RESID) *p = *p + acc[r];
                else *p = acc[r] + bv;
            }
        }
    } else if constexpr (EPI == EPI_PK || EPI == EPI_PK_RELU || EPI == EPI_PK_GELU || EPI == EPI_PK_BIAS || EPI == EPI_PK_GELU_ERF) {
        f32x16 v = acc;
        if constexpr (EPI == EPI_PK_BIAS || EPI == EPI_PK_GELU_ERF || EPI == EPI_PK_GELU) {      // (EPI_PK_GELU: bias optional - the ChemicalOCR tower's fc1)
            static_assert(TOR, "bias epilogues use D = W·X^T (a lane owns a token, its registers are output features)");
            if (a.bias) {          // register 4g + i holds feature n0 + 8g + 4 half + i: one 16-byte load per group (N is a multiple of 4)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + 8 * g + 4 * half;
                    const float4 bv = *(const float4*)(a.bias + (n + 3 < a.N ? n : 0));
                    v[4 * g] += bv.x; v[4 * g + 1] += bv.y; v[4 * g + 2] += bv.z; v[4 * g + 3] += bv.w;
                }
            }
            if constexpr (EPI == EPI_PK_GELU_ERF) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = gelu_erf(v[r]);
            }
        }
        if (EPI == EPI_PK_RELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (EPI == EPI_PK_GELU) {        // torch gelu(approximate="tanh"): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) = x sigmoid(2 u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = v[r], u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
                v[r] = x / (1.0f + fast_exp(-2.0f * u));
            }
        }
        uint4 ch[2];
        acc_to_chunks(v, half, ch);
        const int m = m0 + l32;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n = n0 + 16 * q + 8 * half;
            if (m < a.M && n < a.N && ((qmask >> q) & 1)) st16(a.out_pk + pk_off(m, n, a.N), ch[q]);
        }
    } else {  // EPI_HEADS
        const HeadsOut& ho = a.heads;
        uint4 ch[2];
        acc_to_chunks(acc, half, ch);
        if (TOR) {
            const int m = m0 + l32;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int n = n0 + 16 * q + 8 * half;
                if (m < a.M && n < a.N && ((qmask >> q) & 1)) {
                    const int ri = n / ho.inner, nn = n - ri * ho.inner;
                    heads_store(ho, ri, nn >> 6, m, nn & 63, ch[q]);
                }
            }
        } else {
            const int n = n0 + l32;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int m = m0 + 16 * q + 8 * half;
                if (m < a.M && n < a.N) {
                    const int ri = n / ho.inner, nn = n - ri * ho.inner;
                    heads_store(ho, ri, nn >> 6, m, nn & 63, ch[q]);
                }
            }
        }
    }
}

// EPI_RESID_NORM for one 32-token row tile and a wave's two 32-feature column tiles (D = W·X^T: a lane owns token
// m0 + lane%32 and, per accumulator group g, the 4 consecutive features n0 + 8g + 4*half ..: one float4 of the tiled h).
MG_DEV void resid_norm_epilogue(const GemmArgs& a, const f32x16& acc0, const f32x16& acc1, int m0, int n0, int lane) {
    const int half = lane >> 5, l32 = lane & 31, m = m0 + l32;
    const bool row_ok = m < a.M;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const f32x16& acc = j ? acc1 : acc0;
        const int nj = n0 + 32 * j;
        if (nj >= a.N) continue;                       // (N is a multiple of 32 here: whole tiles only)
        f32x16 xg;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nj + 8 * g + 4 * half;
            float* p = a.out_f32 + ht_off(row_ok ? m : 0, n, a.N);
            float4 hv = *(const float4*)p;
            hv.x += acc[4 * g]; hv.y += acc[4 * g + 1]; hv.z += acc[4 * g + 2]; hv.w += acc[4 * g + 3];
            if (row_ok) { *(float4*)p = hv; ss += (hv.x * hv.x + hv.y * hv.y) + (hv.z * hv.z + hv.w * hv.w); }
            if (a.gain) {
                const float4 gn = *(const float4*)(a.gain + n);
                xg[4 * g] = hv.x * gn.x; xg[4 * g + 1] = hv.y * gn.y; xg[4 * g + 2] = hv.z * gn.z; xg[4 * g + 3] = hv.w * gn.w;
            }
        }
        if (a.gain) {
            uint4 ch[2];
            acc_to_chunks(xg, half, ch);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (row_ok) st16(a.out_pk + pk_off(m, nj + 16 * q + 8 * half, a.N), ch[q]);
        }
    }
    if (a.part) {
        ss += __shfl_xor(ss, 32);
        if (half == 0 && row_ok && n0 < a.N) a.part[(size_t)m * a.ldo + (n0 >> 6)] = ss;      // ldo = partial sums per row
    }
}

// The same update for a wave's TI row tiles, with the residual reads BATCHED.  Written tile by tile as above, hipcc emits every
// 16-byte read of h (and of the gain) followed by s_waitcnt vmcnt(0): a store is pending whenever the next read's value is needed
// and reads and writes share vmcnt on gfx9, so the compiler drains the queue each time - 16 dependent memory round trips per row
// tile, 80 per output tile (the 43 us "epilogue" of profiles/r04_b_gemm_pp_whatif.txt is 80 x ~0.5 us of latency, not bandwidth).
// Here: the 8 reads of a row tile are issued together (untracked, gld16_async) and added INTO the accumulators (which then hold the
// new h: no second copy in registers), the reads of row tile i + 1 are issued before tile i's stores, and one drained wait per row
// tile covers both (reads and writes retire out of order with respect to each other, so a counted wait cannot separate them):
// 5 round trips per output tile.  The gains come from LDS (`gl`: the gains of this wave's 64 columns, staged by the persistent
// kernel once per launch: `gl` = LDS address of the gains of this wave's 64 columns) or, !staged, from global memory.
// Same arithmetic per element (h + acc is commutative) and the same order in the partial sums: bit-identical to resid_norm_epilogue.
template <int TI, bool staged>
MG_DEV void resid_norm_epilogue_tiles(const GemmArgs& a, f32x16 (&acc)[TI][2], const int (&mrow)[TI], int n0, int lane, mg_lds_t gl) {
    const int half = lane >> 5, l32 = lane & 31;
    const bool col_ok[2] = {n0 < a.N, n0 + 32 < a.N};               // (N is a multiple of 32 here: whole tiles only)
    // addresses: a lane's 4 feature groups of a 32-column tile are 1 KiB apart in the tiled h (immediate offsets off one base per
    // (row tile, column tile)); its two 16-byte chunks of the packed output are 1 KiB apart as well
    auto hbase = [&](int i, int j) -> float* {
        const int m = mrow[i] + l32;
        return a.out_f32 + ht_off(m < a.M ? m : 0, (col_ok[j] ? n0 + 32 * j : 0) + 4 * half, a.N);
    };
    mg_raw16 h[8];
    auto issue = [&](int i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* b = hbase(i, j);
            gld16_async_off<0>(h[j * 4 + 0], b);
            gld16_async_off<1024>(h[j * 4 + 1], b);
            gld16_async_off<2048>(h[j * 4 + 2], b);
            gld16_async_off<3072>(h[j * 4 + 3], b);
        }
    };
    issue(0);
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        MG_WAIT_VMCNT_TIE4(0, h[0], h[1], h[2], h[3]);
        MG_TIE(h[4]); MG_TIE(h[5]); MG_TIE(h[6]); MG_TIE(h[7]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint4 r = raw16_get(h[j * 4 + g]);
                f32x16& ac = acc[i][j];
                ac[4 * g] += __uint_as_float(r.x); ac[4 * g + 1] += __uint_as_float(r.y);
                ac[4 * g + 2] += __uint_as_float(r.z); ac[4 * g + 3] += __uint_as_float(r.w);
            }
        MG_TIE(acc[i][0]); MG_TIE(acc[i][1]);          // (the sums exist before the registers of h are handed to the next reads)
        if (i + 1 < TI) issue(i + 1);
        const int m = mrow[i] + l32;
        const bool row_ok = m < a.M;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!col_ok[j]) continue;
            const f32x16& ac = acc[i][j];
            float* const hb = hbase(i, j);
            f32x16 xg;
            const mg_lds_t ga = gl + (32 * j + 4 * half) * 4;       // (staged: gains of columns n0 + 32 j + 4 half + 8 g .. + 3)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 hv = make_float4(ac[4 * g], ac[4 * g + 1], ac[4 * g + 2], ac[4 * g + 3]);
                if (row_ok) { *(float4*)(hb + g * 256) = hv; ss += (hv.x * hv.x + hv.y * hv.y) + (hv.z * hv.z + hv.w * hv.w); }
                if (a.gain) {
                    float4 gv;
                    if (staged) {
                        mg_raw16 gr;
                        if (g == 0) lds_rd16_async<0>(gr, ga); else if (g == 1) lds_rd16_async<32>(gr, ga);
                        else if (g == 2) lds_rd16_async<64>(gr, ga); else lds_rd16_async<96>(gr, ga);
                        MG_WAIT_LGKM_TIE(0, gr);
                        const uint4 r = raw16_get(gr);
                        gv = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
                    } else {
                        gv = *(const float4*)(a.gain + n0 + 32 * j + 8 * g + 4 * half);
                    }
                    xg[4 * g] = hv.x * gv.x; xg[4 * g + 1] = hv.y * gv.y; xg[4 * g + 2] = hv.z * gv.z; xg[4 * g + 3] = hv.w * gv.w;
                }
            }
            if (a.gain) {
                uint4 ch[2];
                acc_to_chunks(xg, half, ch);
                uint16_t* const pb = a.out_pk + pk_off(m, n0 + 32 * j + 8 * half, a.N);
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (row_ok) st16(pb + q * TILE_ELEMS, ch[q]);
            }
        }
        if (a.part) {
            ss += __shfl_xor(ss, 32);
            if (half == 0 && row_ok && col_ok[0]) a.part[(size_t)m * a.ldo + (n0 >> 6)] = ss;      // ldo = partial sums per row
        }
    }
}

MG_DEV bool heads_region_is_T(const HeadsOut& ho, int n) {
    const int ri = n / ho.inner;
    return ho.fmt[ri] == HF_PK_T;
}


// epilogue of a wave's TI x 2 accumulator tiles (token rows mrow[i] .. mrow[i] + 31, feature columns n0w + 32 j) of the large-M kernels
// (GAIN_LDS / gain_lds: EPI_RESID_NORM only, see resid_norm_epilogue_tiles)
template <int EPI, int TI, int XP = 0, bool GAIN_LDS = false>
MG_DEV void xl_epilogue(const GemmArgs& a, f32x16 (&acc)[TI][2], const int (&mrow)[TI], int n0w, bool tor, int lane, mg_lds_t gain_lds = mg_lds_t()) {
    if constexpr (EPI == EPI_RESID_NORM) {
        resid_norm_epilogue_tiles<TI, GAIN_LDS>(a, acc, mrow, n0w, lane, gain_lds);
        return;
    }
    float rsv[TI];
    row_scales_tiles<TI>(a.rs, mrow, a.M, lane, rsv);
    if constexpr (EPI == EPI_HEADS) {          // (the operand order is a property of the whole tile: one branch around the loops)
        if (tor) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) tile_epilogue<EPI_HEADS, true, true>(a, acc[i][j], mrow[i], n0w + 32 * j, lane, 3, rsv[i]);
        } else {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) tile_epilogue<EPI_HEADS, false, true>(a, acc[i][j], mrow[i], n0w + 32 * j, lane, 3, rsv[i]);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m0 = mrow[i], n0 = n0w + 32 * j;
            if constexpr ((XP & 4) != 0) { if ((i || j) && acc[i][j][0] != 123456.789f) continue; }
            if constexpr (EPI == EPI_F32_STORE || EPI == EPI_F32_RESID) {
                tile_epilogue<EPI, false>(a, acc[i][j], m0, n0, lane);
            } else if constexpr (EPI != EPI_RESID_NORM && EPI != EPI_HEADS) {
                tile_epilogue<EPI, true, true>(a, acc[i][j], m0, n0, lane, 3, rsv[i]);
            }
        }
}

constexpr int GX_N = 256, GX_K = 64;          // column tile / K-step of the large-M tile kernels

}  // namespace mg
