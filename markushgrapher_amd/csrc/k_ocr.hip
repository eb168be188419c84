// Element-wise / layout kernels of the ChemicalOCR stage (an Idefics3-class vision-language model, stock
// transformers models/idefics3/modeling_idefics3.py + models/llama/modeling_llama.py).  The contractions run on the GEMM and
// attention kernels of the main path (k_gemm.hip, k_attn.hip, k_decode.hip); what is here is what sits between them:
// LayerNorm, GELU, SwiGLU, rotary embedding + head layouts, pixel shuffle, the image-token merge.  First form (round 2): one
// launch per operation, fp32 intermediates - correctness and a first measurement; the fusions are next-round work.
#include "mg_kernels.h"
#include "mg_ocr.h"

namespace mg {

namespace {

MG_DEV float block_sum(float v, float* red, int tid, int nthreads) {      // all threads get the sum; red: [nthreads / 64]
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (nthreads >> 6); ++i) t += red[i];
    __syncthreads();
    return t;
}

// x_pk[m][0..Kaug) = bf16(LayerNorm(h[m]) * w + b) | 1.0 at column d | 0 ...   (the constant-one column carries the bias of the
// projection that reads x_pk: its packed weight holds the bias in column d).  Optionally h[m] += add_bias (the bias of the
// projection whose result is accumulated into h later in the same sub-layer) and / or out_f32[m] = the normalised row.
__global__ __launch_bounds__(256) void layernorm_pack_kernel(float* h, const float* w, const float* b, const float* add_bias,
                                                             uint16_t* x_pk, float* out_f32, int M, int d, int Kaug, float eps) {
    MG_DYN_SMEM(smem);
    float* red = (float*)smem;                  // [4]
    const int m = blockIdx.x, tid = threadIdx.x;
    float* row = h + (size_t)m * d;
    float s = 0.f;
    for (int i = tid; i < d; i += 256) s += row[i];
    const float mean = block_sum(s, red, tid, 256) / (float)d;
    float v = 0.f;
    for (int i = tid; i < d; i += 256) { const float c = row[i] - mean; v += c * c; }
    const float rstd = rsqrtf(block_sum(v, red, tid, 256) / (float)d + eps);
    for (int i = tid; i < Kaug; i += 256) {
        float y = 0.f;
        if (i < d) {
            const float x = row[i];
            y = (x - mean) * rstd * w[i] + b[i];
            if (out_f32) out_f32[(size_t)m * d + i] = y;
            if (add_bias) row[i] = x + add_bias[i];
        } else if (i == d) {
            y = 1.0f;
        }
        if (x_pk) x_pk[pk_off(m, i, Kaug)] = f32_to_bf16_rn(y);
    }
}

MG_DEV float gelu_tanh(float x) {       // torch gelu(approximate="tanh")
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    return 0.5f * x * (1.0f + tanhf(k0 * (x + k1 * x * x * x)));
}
// y_pk[m][0..Kaug) = bf16(gelu(in[m][0..N))) | 1.0 at column N | 0
__global__ __launch_bounds__(256) void gelu_pack_kernel(const float* in, uint16_t* y_pk, int M, int N, int Kaug) {
    const size_t n = (size_t)M * Kaug;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / Kaug), c = (int)(i - (size_t)m * Kaug);
        const float y = c < N ? gelu_tanh(in[(size_t)m * N + c]) : (c == N ? 1.0f : 0.f);
        y_pk[pk_off(m, c, Kaug)] = f32_to_bf16_rn(y);
    }
}
// y_pk[m][0..I) = bf16(silu(in[m][2c]) * in[m][2c + 1])      (in = gate / up interleaved: the packed weight has gate_j in row 2j and
// up_j in row 2j + 1, so that the decode-step GEMM's SwiGLU epilogue finds a pair in one lane; modeling_llama.py LlamaMLP)
__global__ __launch_bounds__(256) void silu_mul_pack_kernel(const float* in, uint16_t* y_pk, int M, int I) {
    const size_t n = (size_t)M * I;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / I), c = (int)(i - (size_t)m * I);
        const float g = in[(size_t)m * 2 * I + 2 * c], u = in[(size_t)m * 2 * I + 2 * c + 1];
        y_pk[pk_off(m, c, I)] = f32_to_bf16_rn(g / (1.0f + fast_exp(-g)) * u);
    }
}

// hidden[n][p] = patch_emb[n*P + p] + pos_emb[pos_ids[n][p]]   (pos_ids null = full image: position ids = arange,
// modeling_idefics3.py:128-172; otherwise the bucketed fractional coordinates computed by the host wrapper); rows p >= P of an
// image (P_cap > P) are zero.  vmask (nullable): [N][P_cap] key mask of the tower's attention <- patch_mask [N][P] (null = all ones)
__global__ __launch_bounds__(256) void add_pos_kernel(const float* patch, const uint16_t* pos, const int* pos_ids, const uint8_t* patch_mask,
                                                      uint8_t* vmask, float* hidden, int N, int P, int P_cap, int d) {
    if (vmask)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)N * P_cap; i += (size_t)gridDim.x * 256) {
            const int p = (int)(i % P_cap), im = (int)(i / P_cap);
            vmask[i] = p < P ? (patch_mask ? patch_mask[(size_t)im * P + p] != 0 : 1) : 0;
        }
    const size_t n = (size_t)N * P_cap * d;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % d);
        const size_t r = i / d;
        const int p = (int)(r % P_cap), im = (int)(r / P_cap);
        const int pi = (p < P && pos_ids) ? pos_ids[(size_t)im * P + p] : p;
        hidden[i] = p < P ? patch[((size_t)im * P + p) * d + c] + bf16_to_f32(pos[(size_t)pi * d + c]) : 0.f;
    }
}

// Idefics3Connector.pixel_shuffle (modeling_idefics3.py:397-406) + packing: token (n, y2, x2), feature ((dy*sf + dx)*e + c)
// <- vis[n][(y2*sf + dy)*g + x2*sf + dx][c]
__global__ __launch_bounds__(256) void pixel_shuffle_pack_kernel(const float* vis, uint16_t* x_pk, int N, int g, int P_cap, int e, int sf) {
    const int g2 = g / sf, T = g2 * g2, F = e * sf * sf;
    const size_t n = (size_t)N * T * F;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int f = (int)(i % F);
        const size_t r = i / F;
        const int t = (int)(r % T), im = (int)(r / T);
        const int y2 = t / g2, x2 = t - y2 * g2;
        const int c = f % e, q = f / e, dy = q / sf, dx = q - dy * sf;
        const int p = (y2 * sf + dy) * g + x2 * sf + dx;
        x_pk[pk_off((int)r, f, F)] = f32_to_bf16_rn(vis[((size_t)im * P_cap + p) * e + c]);
    }
}

// inputs_merger (modeling_idefics3.py:533-561): h[b][t] = ids[b][t] == image_token ? feats[next image row] : tok_emb[ids[b][t]];
// the k-th <image> position of the batch in row-major order takes feature row k (masked_scatter).  One workgroup per sequence;
// rows t >= L of the padded row space are zero.  err: counts ids outside [0, V) and sequences whose image-token count differs
// from `per_seq`.
__global__ __launch_bounds__(256) void merge_embed_kernel(const int64_t* ids, const uint16_t* tok_emb, const float* feats, float* h, int B,
                                                          int L, int T_cap, int d, int V, int image_token, int per_seq, int* err) {
    MG_DYN_SMEM(smem);
    int* rank = (int*)smem;                     // [2048]
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int c = 0;
        for (int t = 0; t < L; ++t) {
            const bool im = ids[(size_t)b * L + t] == image_token;
            if (t < 2048) rank[t] = im ? c : -1;
            c += im;
        }
        if (c != per_seq && feats) atomicAdd(err, 1);
    }
    __syncthreads();
    for (int i = tid; i < T_cap * d; i += 256) {
        const int t = i / d, c = i - t * d;
        float v = 0.f;
        if (t < L) {
            int64_t id = ids[(size_t)b * L + t];
            if (id < 0 || id >= V) { if (c == 0) atomicAdd(err, 1); id = 0; }
            const int rk = t < 2048 ? rank[t] : -1;
            v = (rk >= 0 && feats && rk < per_seq) ? feats[((size_t)b * per_seq + rk) * d + c] : bf16_to_f32(tok_emb[(size_t)id * d + c]);
        }
        h[((size_t)b * T_cap + t) * d + c] = v;
    }
}

MG_DEV void rope_pair(float& a, float& b, int i, float pos, float theta_log2) {     // dims i and i + 32 of a 64-wide head
    const float inv = exp2f(-(float)(2 * i) / 64.0f * theta_log2);
    float sn, cs;
    sincosf(pos * inv, &sn, &cs);
    const float x = a, y = b;
    a = x * cs - y * sn;
    b = y * cs + x * sn;
}

// cs[pos][0..32) = cos(pos * theta^(-2i/64)), cs[pos][32..64) = sin(...): the rotation table of the rotary attention step
__global__ __launch_bounds__(256) void rope_table_kernel(float* cs, int positions, float theta_log2) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < positions * 32; idx += gridDim.x * 256) {
        const int i = idx & 31, p = idx >> 5;
        const float inv = exp2f(-(float)(2 * i) / 64.0f * theta_log2);
        float sn, c;
        sincosf((float)p * inv, &sn, &c);
        cs[(size_t)p * 64 + i] = c;
        cs[(size_t)p * 64 + 32 + i] = sn;
    }
}

// Prefill: qkv fp32 [B*T_cap][(H + 2*KV)*64] -> rotary embedding (LlamaRotaryEmbedding "default", rotate_half), q * 64^-0.5,
// bf16; written as the attention kernel's operands (Q, K: HF_PK_ROWS; V^T: HF_PK_T, key/value heads repeated H/KV times =
// repeat_kv, which the prefill attention kernel wants materialised) and into the decode caches [B][KV][cap][64] (key/value heads
// once: the decode step's rotary attention shares them among the query heads of a group).  One thread per (row, head, pair i).
__global__ __launch_bounds__(256) void rope_heads_kernel(const float* qkv, int B, int T, int T_cap, int H, int KV, float theta_log2,
                                                         uint16_t* Q, uint16_t* K, uint16_t* Vt, uint16_t* Kc, uint16_t* Vc, int cap) {
    const int rep = H / KV, ld = (H + 2 * KV) * 64;
    const size_t n = (size_t)B * T_cap * H * 32;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
        const int i = (int)(idx & 31);
        const int hh = (int)((idx >> 5) % H);
        const size_t r = (idx >> 5) / H;
        const int t = (int)(r % T_cap), b = (int)(r / T_cap);
        const float* row = qkv + r * ld;
        const int g = hh / rep;
        float q0 = 0.f, q1 = 0.f, k0 = 0.f, k1 = 0.f, v0 = 0.f, v1 = 0.f;
        if (t < T) {
            q0 = row[hh * 64 + i]; q1 = row[hh * 64 + 32 + i];
            k0 = row[(H + g) * 64 + i]; k1 = row[(H + g) * 64 + 32 + i];
            v0 = row[(H + KV + g) * 64 + i]; v1 = row[(H + KV + g) * 64 + 32 + i];
            rope_pair(q0, q1, i, (float)t, theta_log2);
            rope_pair(k0, k1, i, (float)t, theta_log2);
            q0 *= 0.125f; q1 *= 0.125f;
        }
        const uint16_t bq0 = f32_to_bf16_rn(q0), bq1 = f32_to_bf16_rn(q1), bk0 = f32_to_bf16_rn(k0), bk1 = f32_to_bf16_rn(k1);
        const uint16_t bv0 = f32_to_bf16_rn(v0), bv1 = f32_to_bf16_rn(v1);
        // HF_PK_ROWS: [b][h][t/32][dim/16][ (dim/8 & 1)*256 + (t%32)*8 + dim%8 ]
        const size_t rbase = (((size_t)b * H + hh) * (size_t)(T_cap >> 5) + (size_t)(t >> 5)) * (4 * TILE_ELEMS);
        auto rows_off = [&](int dim) { return rbase + (size_t)(dim >> 4) * TILE_ELEMS + (size_t)(((dim >> 3) & 1) * 256 + (t & 31) * 8 + (dim & 7)); };
        Q[rows_off(i)] = bq0; Q[rows_off(i + 32)] = bq1;
        K[rows_off(i)] = bk0; K[rows_off(i + 32)] = bk1;
        // HF_PK_T: [b][h][dim/32][t/16][ (t/8 & 1)*256 + (dim%32)*8 + t%8 ]
        auto t_off = [&](int dim) {
            return ((((size_t)b * H + hh) * 2 + (size_t)(dim >> 5)) * (size_t)(T_cap >> 4) + (size_t)(t >> 4)) * TILE_ELEMS +
                   (size_t)(((t >> 3) & 1) * 256 + (dim & 31) * 8 + (t & 7));
        };
        Vt[t_off(i)] = bv0; Vt[t_off(i + 32)] = bv1;
        if (t < T && hh % rep == 0) {         // decode caches hold the KV heads once: [B][KV][cap][64]
            const size_t c = (((size_t)b * KV + g) * (size_t)cap + (size_t)t) * 64;
            Kc[c + i] = bk0; Kc[c + 32 + i] = bk1; Vc[c + i] = bv0; Vc[c + 32 + i] = bv1;
        }
    }
}
// r(row) = rsqrt(sum(part[row][0..nparts)) * inv_d + eps) of the deferred RMSNorm (RowScale), by the whole workgroup; 1 if part is null
MG_DEV float block_row_scale(const RowScale& rs, int row, float* red, int tid, int nthreads) {
    if (!rs.part) return 1.0f;
    float t = 0.f;
    for (int i = tid; i < rs.nparts; i += nthreads) t += rs.part[(size_t)row * rs.nparts + i];
    return rsqrtf(block_sum(t, red, tid, nthreads) * rs.inv_d + rs.eps);
}
// y_pk[m][0..I) = bf16(silu(r g) * (r u)), r = the deferred RMSNorm scale of row m (gate | up were projected from the un-normalised
// bf16(h * gain)); one workgroup per row
__global__ __launch_bounds__(256) void silu_mul_rows_kernel(const float* in, RowScale rs, uint16_t* y_pk, int M, int I) {
    MG_DYN_SMEM(smem);
    const int m = blockIdx.x, tid = threadIdx.x;
    const float r = block_row_scale(rs, m, (float*)smem, tid, 256);
    for (int c = tid; c < I; c += 256) {
        const float g = in[(size_t)m * 2 * I + 2 * c] * r, u = in[(size_t)m * 2 * I + 2 * c + 1] * r;
        y_pk[pk_off(m, c, I)] = f32_to_bf16_rn(g / (1.0f + fast_exp(-g)) * u);
    }
}
// packed [Npad][K + aug] rows [row0, row0 + N) <- W[N][K] * scale | bias * scale at column K | 0   (rows beyond N up to the next
// multiple of 32 are zeroed when `zero_tail`)
__global__ __launch_bounds__(256) void pack_aug_kernel(const float* W, const float* bias, float scale, uint16_t* dst, int row0, int N, int K,
                                                       int Kaug, int Nfill, int rstride) {
    const size_t n = (size_t)Nfill * Kaug;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / Kaug), c = (int)(i - (size_t)r * Kaug);
        float v = 0.f;
        if (r < N) v = c < K ? W[(size_t)r * K + c] * scale : ((c == K && bias) ? bias[r] * scale : 0.f);
        dst[pk_off(row0 + r * rstride, c, Kaug)] = f32_to_bf16_rn(v);
    }
}

__global__ __launch_bounds__(256) void ocr_init_kernel(int64_t* out_ids, int* unfinished, int* counters, int rows, int max_new, int64_t pad) {
    const int r = blockIdx.x;
    for (int j = threadIdx.x; j < max_new; j += blockDim.x) out_ids[(size_t)r * max_new + j] = pad;
    if (threadIdx.x == 0) {
        unfinished[r] = 1;
        if (r == 0) { counters[0] = rows; counters[1] = -1; counters[2] = 0; counters[5] = 0; counters[6] = 0; }
    }
}
// (lens != null: prompts of different lengths, left-aligned in their rows - row b holds lens[b] <= T tokens)
__global__ __launch_bounds__(256) void last_rows_kernel(int* dst_row, int B, int T, int T_cap, const int* lens) {     // row b*T_cap + T-1 -> b, others dropped
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B * T_cap; i += gridDim.x * 256) {
        const int b = i / T_cap, t = i - b * T_cap;
        dst_row[i] = t == (lens ? lens[b] : T) - 1 ? b : -1;
    }
}
__global__ __launch_bounds__(256) void all_rows_kernel(int* dst_row, int B, int T, int T_cap) {      // row b*T_cap + t -> b*T + t
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B * T_cap; i += gridDim.x * 256) {
        const int b = i / T_cap, t = i - b * T_cap;
        dst_row[i] = t < T ? b * T + t : -1;
    }
}
__global__ __launch_bounds__(256) void key_mask_kernel(uint8_t* mask, int B, int T, int T_cap, const int* lens) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B * T_cap; i += gridDim.x * 256) mask[i] = (i % T_cap) < (lens ? lens[i / T_cap] : T);
}
__global__ __launch_bounds__(256) void len_delta_kernel(const int* lens, int* delta, int N, int L, int* err) {      // delta[n] = lens[n] - L; a length outside [1, L] is reported
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        const int l = lens[i];
        if (l < 1 || l > L) atomicAdd(err, 1);
        delta[i] = (l < 1 ? 1 : (l > L ? L : l)) - L;
    }
}

// row-major [M][d] fp32 <-> the tiled residual layout ht_off (M a multiple of 32): thread = (row of a tile, 4-feature group), so the tiled side moves
// in 512-byte runs
__global__ __launch_bounds__(256) void tile_f32_kernel(const float* src, float* dst, int M, int d, int to_tiled) {
    const size_t n = (size_t)M * (d >> 2);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i & 31);
        const size_t t = i >> 5;
        const int g = (int)(t % (size_t)(d >> 2)), m = (int)(t / (size_t)(d >> 2)) * 32 + r;
        const size_t rm = (size_t)m * d + 4 * g, tl = ht_off(m, 4 * g, d);
        if (to_tiled) *(float4*)(dst + tl) = *(const float4*)(src + rm);
        else *(float4*)(dst + rm) = *(const float4*)(src + tl);
    }
}

int grid_for(size_t n) { const size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g)); }

}  // namespace

void ocr_tile_f32(const float* src, float* dst, int M, int d, int to_tiled, mgStream_t st) {
    MG_LAUNCH(tile_f32_kernel, dim3(grid_for((size_t)M * (d >> 2))), dim3(256), 0, st, src, dst, M, d, to_tiled);
}

void ocr_layernorm_pack(float* h, const float* w, const float* b, const float* add_bias, uint16_t* x_pk, float* out_f32, int M, int d,
                        int Kaug, float eps, mgStream_t st) {
    MG_LAUNCH(layernorm_pack_kernel, dim3(M), dim3(256), 64, st, h, w, b, add_bias, x_pk, out_f32, M, d, Kaug, eps);
}
void ocr_gelu_pack(const float* in, uint16_t* y_pk, int M, int N, int Kaug, mgStream_t st) {
    MG_LAUNCH(gelu_pack_kernel, dim3(grid_for((size_t)M * Kaug)), dim3(256), 0, st, in, y_pk, M, N, Kaug);
}
void ocr_silu_mul_pack(const float* in, uint16_t* y_pk, int M, int I, mgStream_t st) {
    MG_LAUNCH(silu_mul_pack_kernel, dim3(grid_for((size_t)M * I)), dim3(256), 0, st, in, y_pk, M, I);
}
void ocr_add_pos(const float* patch, const uint16_t* pos, const int* pos_ids, const uint8_t* patch_mask, uint8_t* vmask, float* hidden, int N, int P,
                 int P_cap, int d, mgStream_t st) {
    MG_LAUNCH(add_pos_kernel, dim3(grid_for((size_t)N * P_cap * d)), dim3(256), 0, st, patch, pos, pos_ids, patch_mask, vmask, hidden, N, P, P_cap, d);
}
void ocr_pixel_shuffle_pack(const float* vis, uint16_t* x_pk, int N, int g, int P_cap, int e, int sf, mgStream_t st) {
    MG_LAUNCH(pixel_shuffle_pack_kernel, dim3(grid_for((size_t)N * (g / sf) * (g / sf) * e * sf * sf)), dim3(256), 0, st, vis, x_pk, N, g, P_cap, e, sf);
}
void ocr_merge_embed(const int64_t* ids, const uint16_t* tok_emb, const float* feats, float* h, int B, int L, int T_cap, int d, int V,
                     int image_token, int per_seq, int* err, mgStream_t st) {
    MG_LAUNCH(merge_embed_kernel, dim3(B), dim3(256), 2048 * sizeof(int), st, ids, tok_emb, feats, h, B, L, T_cap, d, V, image_token, per_seq, err);
}
void ocr_rope_heads(const float* qkv, int B, int T, int T_cap, int H, int KV, float theta, uint16_t* Q, uint16_t* K, uint16_t* Vt,
                    uint16_t* Kc, uint16_t* Vc, int cap, mgStream_t st) {
    MG_LAUNCH(rope_heads_kernel, dim3(grid_for((size_t)B * T_cap * H * 32)), dim3(256), 0, st, qkv, B, T, T_cap, H, KV, log2f(theta), Q, K, Vt, Kc, Vc, cap);
}
void ocr_rope_table(float* cs, int positions, float theta, mgStream_t st) {
    MG_LAUNCH(rope_table_kernel, dim3(grid_for((size_t)positions * 32)), dim3(256), 0, st, cs, positions, log2f(theta));
}
void ocr_silu_mul_rows(const float* in, const RowScale& rs, uint16_t* y_pk, int M, int I, mgStream_t st) {
    MG_LAUNCH(silu_mul_rows_kernel, dim3(M), dim3(256), 64, st, in, rs, y_pk, M, I);
}
void ocr_pack_aug(const float* W, const float* bias, float scale, uint16_t* dst, int row0, int N, int K, int Kaug, int Nfill, int rstride, mgStream_t st) {
    MG_LAUNCH(pack_aug_kernel, dim3(grid_for((size_t)Nfill * Kaug)), dim3(256), 0, st, W, bias, scale, dst, row0, N, K, Kaug, Nfill, rstride);
}
void ocr_init(int64_t* out_ids, int* unfinished, int* counters, int rows, int max_new, int64_t pad, mgStream_t st) {
    MG_LAUNCH(ocr_init_kernel, dim3(rows), dim3(256), 0, st, out_ids, unfinished, counters, rows, max_new, pad);
}
__global__ __launch_bounds__(256) void ocr_slots_init_kernel(int* unfinished, int* pos, int* img, int* pool, int64_t* next_ids, int slots, int* ctr, int N) {
    for (int r = threadIdx.x; r < slots; r += blockDim.x) { unfinished[r] = 0; pos[r] = 0; img[r] = -1; pool[r] = 0; next_ids[r] = 0; }
    if (threadIdx.x == 0) ctr[8] = N;
}
__global__ __launch_bounds__(256) void ocr_fill_ints_kernel(int* p, int v, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = v;
}
__global__ void ocr_add_int_kernel(int* dst, const int* src) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst += *src; }
__global__ void ocr_set_int_kernel(int* dst, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v; }
void ocr_slots_init(int* unfinished, int* pos, int* img, int* pool, int64_t* next_ids, int slots, int* ctr, int N, mgStream_t st) {
    MG_LAUNCH(ocr_slots_init_kernel, dim3(1), dim3(256), 0, st, unfinished, pos, img, pool, next_ids, slots, ctr, N);
}
void ocr_fill_ints(int* p, int v, int n, mgStream_t st) { MG_LAUNCH(ocr_fill_ints_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, st, p, v, n); }
void ocr_add_int(int* dst, const int* src, mgStream_t st) { MG_LAUNCH(ocr_add_int_kernel, dim3(1), dim3(64), 0, st, dst, src); }
void ocr_set_int(int* dst, int v, mgStream_t st) { MG_LAUNCH(ocr_set_int_kernel, dim3(1), dim3(64), 0, st, dst, v); }
void ocr_row_maps(int* last_rows, int* all_rows, uint8_t* key_mask, int B, int T, int T_cap, mgStream_t st, const int* lens) {
    const int g = grid_for((size_t)B * T_cap);
    MG_LAUNCH(last_rows_kernel, dim3(g), dim3(256), 0, st, last_rows, B, T, T_cap, lens);
    MG_LAUNCH(all_rows_kernel, dim3(g), dim3(256), 0, st, all_rows, B, T, T_cap);
    MG_LAUNCH(key_mask_kernel, dim3(g), dim3(256), 0, st, key_mask, B, T, T_cap, lens);
}
void ocr_len_delta(const int* lens, int* delta, int N, int L, int* err, mgStream_t st) {
    MG_LAUNCH(len_delta_kernel, dim3(grid_for((size_t)N)), dim3(256), 0, st, lens, delta, N, L, err);
}

}  // namespace mg
