// Kernels of the OCSR vision branch "e1" (SURVEY.md §8 rows a7 / f-2): a Swin encoder as stock transformers
// models/swin/modeling_swin.py states it (the importable upstream of MolScribe's timm swin_base_patch4_window12_384, which the
// reference loads into model.encoder.molscribe_encoder: ref markushgrapher/core/common/begin.py:137-138).  The contractions run on
// the GEMM kernels of the main path (k_gemm.hip / k_gemm_pp.hip, bias and exact-GELU epilogues added for this branch); what is here
// is what sits between them: input derivation, im2col, LayerNorm (+ patch-merging gather), and the (shifted-)window attention.
//
// Window attention on gfx950: a window is n = w*w tokens (144 for w = 12) of head dim 32 - 9 tiles of v_mfma_f32_16x16x32_bf16,
// whose K = 32 is exactly one head.  One wave owns one (image, window, head): K fragments stay in registers, V sits transposed in
// LDS, the scores are computed TRANSPOSED (S^T = K Q^T: a lane owns one query column and 4 keys per tile), so that the softmax
// reduction over keys is in-lane plus two cross-lane steps, and the rounded weights are ALREADY the B operand of the second product
// out^T = V^T P^T when the MFMA's k index is read as the key order the lanes happen to hold (the sum over keys does not care about
// their order, V^T is read from LDS in the same order).  Relative-position bias, shift mask and the cyclic shift itself are index
// arithmetic inside the kernel: neither the [heads][n][n] bias nor the [windows][n][n] mask is ever materialised, the rolled /
// partitioned copies of the activations that stock makes (torch.roll, window_partition, window_reverse) do not exist.
#include "mg_swin.h"

namespace mg {

namespace {
int grid_for(size_t n) { const size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g)); }
}  // namespace
// (the kernels below have external names so that kernel traces show them)

// ---------------------------------------------------------------------------------------------------------
// input derivation (INFERRED piece of the fork: see e1_shapes.py) and im2col
// ---------------------------------------------------------------------------------------------------------
// torch upsample_bilinear2d, align_corners = false (aten UpSample.h area_pixel_compute_source_index / guard_index_and_lambda):
// src = max(scale * (dst + 0.5) - 0.5, 0), i0 = min(int(src), S - 1), i1 = min(i0 + 1, S - 1), l1 = src - i0, l0 = 1 - l1
__global__ __launch_bounds__(256) void swin_resize_kernel(const float* src, float* dst, int B, int C, int S, int I, SwinPixAffine af) {
    const size_t n = (size_t)B * C * I * I;
    const float sc = (float)S / (float)I;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % I), y = (int)((i / I) % I);
        const size_t bc = i / ((size_t)I * I);
        const int c = (int)(bc % C);
        float fy = sc * ((float)y + 0.5f) - 0.5f, fx = sc * ((float)x + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
        int y0 = (int)fy, x0 = (int)fx;
        y0 = y0 < S - 1 ? y0 : S - 1; x0 = x0 < S - 1 ? x0 : S - 1;
        const int y1 = y0 + 1 < S - 1 ? y0 + 1 : S - 1, x1 = x0 + 1 < S - 1 ? x0 + 1 : S - 1;
        float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
        ly1 = ly1 < 0.f ? 0.f : (ly1 > 1.f ? 1.f : ly1); lx1 = lx1 < 0.f ? 0.f : (lx1 > 1.f ? 1.f : lx1);
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float* p = src + bc * (size_t)S * S;
        const float v = ly0 * (lx0 * p[(size_t)y0 * S + x0] + lx1 * p[(size_t)y0 * S + x1]) +
                        ly1 * (lx0 * p[(size_t)y1 * S + x0] + lx1 * p[(size_t)y1 * S + x1]);
        dst[i] = v * af.scale[c] + af.shift[c];
    }
}

__global__ __launch_bounds__(256) void swin_im2col_pack_kernel(const float* pix, uint16_t* x_pk, int B, int C, int I, int ps, int Kp) {
    const int n = I / ps, P = n * n, K = C * ps * ps, M = B * P;
    const size_t nchunk = (size_t)((M + 31) >> 5) * (size_t)(Kp >> 4) * 64;
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < nchunk; c += (size_t)gridDim.x * 256) {
        const size_t tile = c >> 6;
        const int l = (int)(c & 63);
        const int rt = (int)(tile / (size_t)(Kp >> 4)), kt = (int)(tile % (size_t)(Kp >> 4));
        const int m = rt * 32 + (l & 31), k0 = kt * 16 + 8 * (l >> 5);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            v[j] = 0.f;
            if (m < M && k < K) {
                const int b = m / P, p = m - b * P, py = p / n, px = p - py * n;
                const int ch = k / (ps * ps), kk = k - ch * ps * ps, ky = kk / ps, kx = kk - ky * ps;
                v[j] = pix[(((size_t)b * C + ch) * I + (size_t)(py * ps + ky)) * I + (size_t)(px * ps + kx)];
            }
        }
        st16(x_pk + c * 8, make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])));
    }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm (+ patch-merging gather) over the TILED fp32 residual stream (ht_off: [rows/32][C/4][32 rows][4 features], the layout
// the encoder GEMMs' residual epilogue reads and writes in 16-byte groups) -> packed bf16 / fp32.
// A workgroup takes one 32-row tile: thread (row r = tid % 32, q = tid / 32) owns the 4-feature groups q, q + 8, ...: the 32 threads
// of a q read 512 contiguous bytes per group.  Row statistics: in-thread sums, the two q of a wave by one xor-32 exchange, the four
// waves through LDS in a fixed order.  KEEP: the row's values stay in registers (C <= 1024); else three passes over L2 (the merges).
// ---------------------------------------------------------------------------------------------------------
template <int GPT, bool KEEP>
__global__ __launch_bounds__(256) void swin_ln_kernel(SwinLnArgs a) {
    MG_DYN_SMEM(smem);
    float (*red)[4][32] = (float (*)[4][32])smem;          // [2][4][32]
    const int tid = threadIdx.x, r = tid & 31, q = tid >> 5, wv = tid >> 6;
    const int T = blockIdx.x, m = 32 * T + r, C = a.C;
    const bool ok = m < a.M;
    const int mm = ok ? m : a.M - 1;
    // source of feature group g (4 features) of this thread's row
    const int Cin = a.merge_R > 0 ? (C >> 2) : C;
    int mb = 0, oi = 0, oj = 0;
    if (a.merge_R > 0) { const int R2 = a.merge_R >> 1; mb = mm / (R2 * R2); const int rr = mm - mb * R2 * R2; oi = rr / R2; oj = rr - oi * R2; }
    auto src = [&](int g) -> const float* {
        const int n = 4 * g;
        if (a.merge_R > 0) {        // stock:318-320: [row::2, col::2] for col in (0, 1) for row in (0, 1), concatenated on the feature axis
            const int part = n / Cin, cin = n - part * Cin, R = a.merge_R;
            const int ms = (mb * R + 2 * oi + (part & 1)) * R + 2 * oj + (part >> 1);
            return a.h_in + ht_off(ms, cin, Cin);
        }
        return a.in_tiled ? a.h_in + ht_off(mm, n, C) : a.h_in + (size_t)mm * C + n;
    };
    float4 v[KEEP ? GPT : 1];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        const float4 x = *(const float4*)src(q + 8 * i);
        if (KEEP) v[i] = x;
        s += (x.x + x.y) + (x.z + x.w);
    }
    s += __shfl_xor(s, 32);
    if ((tid & 32) == 0) red[0][wv][r] = s;
    __syncthreads();
    const float mean = (((red[0][0][r] + red[0][1][r]) + red[0][2][r]) + red[0][3][r]) / (float)C;
    float qv = 0.f;
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        const float4 x = KEEP ? v[KEEP ? i : 0] : *(const float4*)src(q + 8 * i);
        const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
        qv += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    qv += __shfl_xor(qv, 32);
    if ((tid & 32) == 0) red[1][wv][r] = qv;
    __syncthreads();
    const float rstd = rsqrtf((((red[1][0][r] + red[1][1][r]) + red[1][2][r]) + red[1][3][r]) / (float)C + a.eps);
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        const int g = q + 8 * i, n = 4 * g;
        const float4 x = KEEP ? v[KEEP ? i : 0] : *(const float4*)src(g);
        const float4 wg = *(const float4*)(a.w + n), bg = *(const float4*)(a.b + n);
        const float y0 = (x.x - mean) * rstd * wg.x + bg.x, y1 = (x.y - mean) * rstd * wg.y + bg.y;
        const float y2 = (x.z - mean) * rstd * wg.z + bg.z, y3 = (x.w - mean) * rstd * wg.w + bg.w;
        if (a.h_out) {                                     // tiled copy of the row (+ the bias of the projection accumulated into it later)
            float4 o = a.h_out_norm ? make_float4(y0, y1, y2, y3) : x;
            if (a.add_bias) { const float4 ab = *(const float4*)(a.add_bias + n); o.x += ab.x; o.y += ab.y; o.z += ab.z; o.w += ab.w; }
            *(float4*)(a.h_out + ht_off(m, n, C)) = o;
        }
        if (a.out_f32) *(float4*)(a.out_f32 + (size_t)m * C + n) = make_float4(y0, y1, y2, y3);
        if (a.x_pk) *(uint2*)(a.x_pk + pk_off(m, n, a.kaug ? a.kaug : C)) = make_uint2(pack_bf16(y0, y1), pack_bf16(y2, y3));
    }
    // kaug > C (the ChemicalOCR tower's projections carry their bias in a constant-one column, ocr.hip): columns C .. kaug - 1 = [1, 0, 0, ...]
    if (a.x_pk && a.kaug > C)
        for (int c8 = q; 8 * c8 < a.kaug - C; c8 += 8)
            st16(a.x_pk + pk_off(m, C + 8 * c8, a.kaug), make_uint4(c8 == 0 ? 0x00003F80u : 0u, 0u, 0u, 0u));
}

// ---------------------------------------------------------------------------------------------------------
// (shifted-)window attention, head dim 32, NT = w*w / 16 key / query tiles
// ---------------------------------------------------------------------------------------------------------
constexpr int SW_HPB = 4;                               // heads (waves) per workgroup

template <int NT>
struct SwinAttnSmem {
    static constexpr int N = 16 * NT;                   // tokens of a window
    static constexpr int NS = (NT + 1) / 2;             // 32-key steps of the second product
    static constexpr int NPAD = 32 * NS + 8;            // keys per V^T row in LDS (+8: rows 16 B apart modulo the bank period)
    static constexpr int INFO_BYTES = N * 2 * (int)sizeof(int);                 // kinfo | rowidx
    static constexpr int VT_BYTES = 32 * NPAD * 2;
    static int table_bytes(int w) { return ((2 * w - 1) * (2 * w - 1) * (int)sizeof(float) + 15) & ~15; }
    static int bytes(int w, int waves) { return INFO_BYTES + waves * (VT_BYTES + table_bytes(w)); }
};

template <int NT>
__global__ __launch_bounds__(64 * SW_HPB) void swin_attn_kernel(SwinAttnArgs a, int hpb, int tab_bytes) {
    using SM = SwinAttnSmem<NT>;
    constexpr int N = SM::N, NS = SM::NS, NPAD = SM::NPAD;
    MG_DYN_SMEM(smem);
    int* kinfo = (int*)smem;                             // [N]: (iy * (2w-1) + ix) | region id << 16
    int* rowidx = kinfo + N;                             // [N]: row of the token in the natural order
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int w = a.w, R = a.R, nw = R / w, C = a.C, ld = 3 * C;
    const int hgroups = a.H / hpb;
    int bid = blockIdx.x;
    const int hg = bid % hgroups; bid /= hgroups;
    const int win = bid % (nw * nw), b = bid / (nw * nw);
    const int wy = win / nw, wx = win - wy * nw;
    const int tw = 2 * w - 1;
    for (int t = tid; t < N; t += blockDim.x) {
        const int iy = t / w, ix = t - iy * w;
        const int hy = wy * w + iy, hx = wx * w + ix;                 // position in the rolled map
        int oy = hy + a.shift, ox = hx + a.shift;                     // torch.roll(x, -shift): rolled[h] = x[(h + shift) mod R]
        oy = oy >= R ? oy - R : oy; ox = ox >= R ? ox - R : ox;
        int rid = 0;
        if (a.shift > 0) {                                            // stock:584-607 on the rolled coordinates
            const int ry = (hy >= R - w ? 1 : 0) + (hy >= R - a.shift ? 1 : 0), rx = (hx >= R - w ? 1 : 0) + (hx >= R - a.shift ? 1 : 0);
            rid = ry * 3 + rx;
        }
        kinfo[t] = (iy * tw + ix) | (rid << 16);
        rowidx[t] = (b * R + oy) * R + ox;
    }
    char* wbase = smem + SM::INFO_BYTES + wv * (SM::VT_BYTES + tab_bytes);
    uint16_t* vt = (uint16_t*)wbase;                     // [32 dims][NPAD keys]
    float* tab = (float*)(wbase + SM::VT_BYTES);         // [(2w-1)^2] of this wave's head
    __syncthreads();
    // (a wave beyond the stage's head count - fewer than SW_HPB heads - repeats the last head and stores nothing: every wave reaches
    // the barrier below)
    const bool active = wv < hpb;
    const int hd = hg * hpb + (active ? wv : hpb - 1);
    const int l16 = lane & 15, g = lane >> 4;
    // the head's bias table and V^T (zero beyond the window's keys) into this wave's LDS
    for (int i = lane; i < tw * tw; i += 64) tab[i] = a.table[(size_t)hd * tw * tw + i];
    for (int i = lane; i < 32 * (NPAD - N) ; i += 64) { const int dim = i / (NPAD - N), k = N + i % (NPAD - N); vt[dim * NPAD + k] = 0; }
    for (int i = lane; i < N * 4; i += 64) {
        const int t = i >> 2, c8 = (i & 3) * 8;
        const uint4 ch = ld16(a.qkv + pk_off(rowidx[t], 2 * C + hd * 32 + c8, ld));
        const uint32_t wds[4] = {ch.x, ch.y, ch.z, ch.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vt[(c8 + 2 * j) * NPAD + t] = (uint16_t)(wds[j] & 0xFFFFu);
            vt[(c8 + 2 * j + 1) * NPAD + t] = (uint16_t)(wds[j] >> 16);
        }
    }
    // K fragments: A operand of S^T = K Q^T - lane (key 16 kt + l16, dims 8g .. 8g+7)
    uint4 kf[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) kf[kt] = ld16(a.qkv + pk_off(rowidx[16 * kt + l16], C + hd * 32 + 8 * g, ld));
    __syncthreads();                                     // the wave's V^T and table are complete in LDS
    // V^T fragments: A operand of out^T = V^T P^T for step s, dims 16 dt + l16: MFMA k index 8g + i <-> key 32s + 4g + i (i < 4),
    // 32s + 16 + 4g + i - 4 (i >= 4) - the order in which the lanes hold the scores of two consecutive key tiles
    uint4 vf[NS][2];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const uint16_t* r = vt + (16 * dt + l16) * NPAD + 32 * s + 4 * g;
            const uint2 lo = *(const uint2*)r, hi = *(const uint2*)(r + 16);
            vf[s][dt] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    const float scale = 0.17677669529663687f;             // 32^-0.5 (stock:408)
    const int code0 = (w - 1) * tw + (w - 1);
    for (int qt = 0; qt < NT; ++qt) {
        const int tq = 16 * qt + l16;
        const int qrow = rowidx[tq];
        const uint4 qf = ld16(a.qkv + pk_off(qrow, hd * 32 + 8 * g, ld));
        const int qi = kinfo[tq], qcode = (qi & 0xFFFF) + code0, qrid = qi >> 16;
        float sc[NT][4];
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const f32x4 acc = mfma16(kf[kt], qf, acc4_zero());          // acc[j] = k(16 kt + 4g + j) . q(tq)
            const uint4 ki = *(const uint4*)(kinfo + 16 * kt + 4 * g);
            const int kin[4] = {(int)ki.x, (int)ki.y, (int)ki.z, (int)ki.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[j] * scale + tab[qcode - (kin[j] & 0xFFFF)];
                v += ((kin[j] >> 16) != qrid) ? -100.0f : 0.f;
                sc[kt][j] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
        uint32_t pk[NT][2];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            float p[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = fast_exp(sc[kt][j] - mx);
            pk[kt][0] = pack_bf16(p[0], p[1]);
            pk[kt][1] = pack_bf16(p[2], p[3]);
            sum += (bf16lo(pk[kt][0]) + bf16hi(pk[kt][0])) + (bf16lo(pk[kt][1]) + bf16hi(pk[kt][1]));     // the ROUNDED weights sum to the divisor
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        f32x4 o[2] = {acc4_zero(), acc4_zero()};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bool two = 2 * s + 1 < NT;
            const uint4 pf = make_uint4(pk[2 * s][0], pk[2 * s][1], two ? pk[2 * s + 1 < NT ? 2 * s + 1 : 0][0] : 0u,
                                        two ? pk[2 * s + 1 < NT ? 2 * s + 1 : 0][1] : 0u);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) o[dt] = mfma16(vf[s][dt], pf, o[dt]);      // o[dt][j] = out(tq)[16 dt + 4g + j]
        }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const uint2 st = make_uint2(pack_bf16(o[dt][0] * inv, o[dt][1] * inv), pack_bf16(o[dt][2] * inv, o[dt][3] * inv));
            if (active) *(uint2*)(a.ctx + pk_off(qrow, hd * 32 + 16 * dt + 4 * g, C)) = st;
        }
    }
}

__global__ __launch_bounds__(256) void swin_transpose_kernel(const float* src, float* dst, int n, int H) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n * H; i += gridDim.x * 256) { const int h = i / n, j = i - h * n; dst[i] = src[(size_t)j * H + h]; }
}

void swin_resize(const float* src, float* dst, int B, int C, int S, int I, const SwinPixAffine& af, mgStream_t st) {
    MG_LAUNCH(swin_resize_kernel, dim3(grid_for((size_t)B * C * I * I)), dim3(256), 0, st, src, dst, B, C, S, I, af);
}
void swin_im2col_pack(const float* pix, uint16_t* x_pk, int B, int C, int I, int ps, int Kp, mgStream_t st) {
    const int n = I / ps, M = B * n * n;
    const size_t nchunk = (size_t)((M + 31) >> 5) * (size_t)(Kp >> 4) * 64;
    MG_LAUNCH(swin_im2col_pack_kernel, dim3(grid_for(nchunk)), dim3(256), 0, st, pix, x_pk, B, C, I, ps, Kp);
}

bool swin_ln_supported(int C) { return C == 64 || C == 128 || C == 256 || C == 512 || C == 1024 || C == 2048 || C == 4096; }
void swin_layernorm(const SwinLnArgs& a, mgStream_t st) {
#define MG_SWIN_LN(GPT, KEEP) MG_LAUNCH((swin_ln_kernel<GPT, KEEP>), dim3((a.M + 31) / 32), dim3(256), 2 * 4 * 32 * sizeof(float), st, a)
    switch (a.C) {
        case 64: MG_SWIN_LN(2, true); break;
        case 128: MG_SWIN_LN(4, true); break;
        case 256: MG_SWIN_LN(8, true); break;
        case 512: MG_SWIN_LN(16, true); break;
        case 768: MG_SWIN_LN(24, true); break;           // (SigLIP-base width: the ChemicalOCR vision tower, ocr.hip)
        case 1024: MG_SWIN_LN(32, true); break;
        case 2048: MG_SWIN_LN(64, false); break;
        default: MG_SWIN_LN(128, false); break;
    }
#undef MG_SWIN_LN
}

bool swin_attention_supported(int w, int R, int C, int H) {
    return (w == 4 || w == 8 || w == 12) && R % w == 0 && C == 32 * H && (H % SW_HPB == 0 || H < SW_HPB);
}
void swin_attention(const SwinAttnArgs& a, mgStream_t st) {
    const int hpb = a.H < SW_HPB ? a.H : SW_HPB;
    const int nw = a.R / a.w;
    const dim3 grid(a.B * nw * nw * (a.H / hpb)), block(64 * SW_HPB);
#define MG_SWIN_ATT(NT) { const int tb = SwinAttnSmem<NT>::table_bytes(a.w); const size_t sh = SwinAttnSmem<NT>::bytes(a.w, SW_HPB); \
                          static bool once = false; if (!once) { MG_SET_MAX_SMEM((&swin_attn_kernel<NT>), sh); once = true; } \
                          MG_LAUNCH((swin_attn_kernel<NT>), grid, block, sh, st, a, hpb, tb); }
    if (a.w == 4) MG_SWIN_ATT(1)
    else if (a.w == 8) MG_SWIN_ATT(4)
    else MG_SWIN_ATT(9)
#undef MG_SWIN_ATT
}

void swin_transpose_f32(const float* src, float* dst, int n, int H, mgStream_t st) {
    MG_LAUNCH(swin_transpose_kernel, dim3(grid_for((size_t)n * H)), dim3(256), 0, st, src, dst, n, H);
}

}  // namespace mg
