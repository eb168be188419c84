// Encoder input assembly = combine_image_text_embeddings + get_visual_bbox + UdopCellEmbeddings
// (stock:135-251, 814-840), batched on the device instead of per-image Python loops.
//
// Reproduced quirks (verified against stock in tools/make_golden.py):
//  * the patch under EVERY text token is dropped from the visual list, also for tokens whose box mean is 0 or 1
//    (question/pad -> patch 0, sep -> last patch) although nothing is added to those tokens (stock:201-218);
//  * surviving patches keep raster order; the visual list is padded back to P with zero embeddings, box 0,
//    mask 0 (mask 1 when the caller passed no attention_mask: stock:1183-1186);
//  * bbox is float64 after the combine (stock:200): cell indices and box centres are computed in float64,
//    the OCR patch index in float32 (stock:191-198).
#include "mg_kernels.h"

namespace mg {

struct EmbedMeta { int tok; int patch; int c[4]; };   // per (b, s): token id or -1, patch row or -1, cell indices

MG_DEV int cell_index(double v, int M2) {
    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    int i = (int)(v * (double)(M2 - 1));
    return i < 0 ? 0 : (i > M2 - 1 ? M2 - 1 : i);
}

// one workgroup per image: flags, prefix sums, per-token metadata
__global__ __launch_bounds__(256) void embed_index_kernel(EmbedArgs a, EmbedMeta* meta) {
    MG_DYN_SMEM(smem);
    const int tid = threadIdx.x, b = blockIdx.x;
    const int L = a.L, P = a.P, S = L + P, n = a.n_side;
    unsigned char* drop = (unsigned char*)smem;                 // [P]
    int* cnt = (int*)(smem + ((P + 15) & ~15));                 // [256]
    EmbedMeta* mb = meta + (size_t)b * a.S_cap;
    const float* bb = a.bbox + (size_t)b * L * 4;

    for (int p = tid; p < P; p += 256) drop[p] = 0;
    // per-image text length: index of the last attended text token + 1 (trim_padding; all of L otherwise)
    int* lbs = cnt;
    if (tid == 0) lbs[0] = 0;
    __syncthreads();
    if (a.trim_padding && a.attn_mask) {
        int mine = 0;
        for (int i = tid; i < L; i += 256) if (a.attn_mask[(size_t)b * L + i] != 0) mine = i + 1;
        atomicMax(lbs, mine);
    }
    __syncthreads();
    const int Lb = (a.trim_padding && a.attn_mask) ? lbs[0] : L;
    __syncthreads();
    if (tid == 0 && a.text_len) a.text_len[b] = Lb;
    // padded text slots behind the image's own text (trim_padding): moved to the end of the sequence, masked, no effect on the patches
    for (int i = Lb + tid; i < L; i += 256) {
        const int s = Lb + P + (i - Lb);
        EmbedMeta m;
        m.tok = 0; m.patch = -1; m.c[0] = m.c[1] = m.c[2] = m.c[3] = 0;
        mb[s] = m;
        a.cx[(size_t)b * a.S_cap + s] = 0.0;
        a.cy[(size_t)b * a.S_cap + s] = 0.0;
        a.mask[(size_t)b * a.S_cap + s] = 0;
    }
    for (int i = tid; i < Lb; i += 256) {
        const float x0 = bb[i * 4 + 0], y0 = bb[i * 4 + 1], x1 = bb[i * 4 + 2], y1 = bb[i * 4 + 3];
        // stock:191-198 (float32 arithmetic, floor, clip)
        int px = (int)floorf((x0 + x1) / 2.0f * (float)n);
        int py = (int)floorf((y0 + y1) / 2.0f * (float)n);
        px = px < 0 ? 0 : (px > n - 1 ? n - 1 : px);
        py = py < 0 ? 0 : (py > n - 1 ? n - 1 : py);
        const int pt = px + py * n;
        drop[pt] = 1;   // benign race: every writer stores 1
        // stock:200-205 (float64 mean == 0 or == 1 -> nothing added)
        const double mean = ((double)x0 + (double)y0 + (double)x1 + (double)y1) / 4.0;
        const bool seg = (mean == 0.0) || (mean == 1.0);
        long long id = a.input_ids[(size_t)b * L + i];
        if (id < 0 || id >= a.V) { atomicAdd(a.err, 1); id = 0; }
        EmbedMeta m;
        m.tok = (int)id;
        m.patch = seg ? -1 : b * P + pt;
        m.c[0] = cell_index((double)x0, a.M2); m.c[1] = cell_index((double)y0, a.M2);
        m.c[2] = cell_index((double)x1, a.M2); m.c[3] = cell_index((double)y1, a.M2);
        mb[i] = m;
        a.cx[(size_t)b * a.S_cap + i] = ((double)x0 + (double)x1) / 2.0;
        a.cy[(size_t)b * a.S_cap + i] = ((double)y0 + (double)y1) / 2.0;
        a.mask[(size_t)b * a.S_cap + i] = a.attn_mask ? (a.attn_mask[(size_t)b * L + i] != 0) : 1;
    }
    __syncthreads();
    // exclusive prefix sum of "kept" over patches, in raster order
    const int per = (P + 255) / 256;
    int c = 0;
    for (int p = tid * per; p < P && p < (tid + 1) * per; ++p) c += drop[p] ? 0 : 1;
    cnt[tid] = c;
    __syncthreads();
    int off = 0;
    for (int t = 0; t < tid; ++t) off += cnt[t];
    int nsurv = 0;
    for (int t = 0; t < 256; ++t) nsurv += cnt[t];
    for (int p = tid * per; p < P && p < (tid + 1) * per; ++p) {
        if (drop[p]) continue;
        const int s = Lb + off++;
        // stock:135-155 visual boxes (float32 k/n, promoted to float64 by the concat at stock:248)
        const int px = p % n, py = p / n;
        const float vx0 = (float)px / (float)n, vx1 = (float)(px + 1) / (float)n;
        const float vy0 = (float)py / (float)n, vy1 = (float)(py + 1) / (float)n;
        EmbedMeta m;
        m.tok = -1;
        m.patch = b * P + p;
        m.c[0] = cell_index((double)vx0, a.M2); m.c[1] = cell_index((double)vy0, a.M2);
        m.c[2] = cell_index((double)vx1, a.M2); m.c[3] = cell_index((double)vy1, a.M2);
        mb[s] = m;
        a.cx[(size_t)b * a.S_cap + s] = ((double)vx0 + (double)vx1) / 2.0;
        a.cy[(size_t)b * a.S_cap + s] = ((double)vy0 + (double)vy1) / 2.0;
        a.mask[(size_t)b * a.S_cap + s] = 1;
    }
    // zero-padded visual slots (box 0) and the internal padding up to S_cap
    for (int s = Lb + nsurv + tid; s < a.S_cap; s += 256) {
        if (s >= Lb + P && s < S) continue;                    // the moved text padding (written above)
        EmbedMeta m;
        m.tok = -1; m.patch = -1;
        m.c[0] = m.c[1] = m.c[2] = m.c[3] = (s < S) ? 0 : -1;
        mb[s] = m;
        a.cx[(size_t)b * a.S_cap + s] = 0.0;
        a.cy[(size_t)b * a.S_cap + s] = 0.0;
        a.mask[(size_t)b * a.S_cap + s] = (s < S && a.attn_mask == nullptr) ? 1 : 0;
    }
    __syncthreads();
    // compaction map for the cross-attention K/V streams (cross-attention has no positional term, so masked
    // encoder positions are simply not stored): xrow[s] = rank of s among attended positions
    const int per2 = (a.S_cap + 255) / 256;
    c = 0;
    for (int s = tid * per2; s < a.S_cap && s < (tid + 1) * per2; ++s) c += a.mask[(size_t)b * a.S_cap + s] ? 1 : 0;
    cnt[tid] = c;
    __syncthreads();
    off = 0;
    for (int t = 0; t < tid; ++t) off += cnt[t];
    for (int s = tid * per2; s < a.S_cap && s < (tid + 1) * per2; ++s)
        a.xrow[(size_t)b * a.S_cap + s] = a.mask[(size_t)b * a.S_cap + s] ? a.x_row0 + off++ : -1;
    if (tid == 255) a.xlen[b] = a.x_row0 + off;
}

// one wave per output row: hidden = (tok + patch) + (((x[l] + y[u]) + x[r]) + y[lo])   (stock:206, 833-838, 1162)
__global__ __launch_bounds__(256) void embed_gather_kernel(EmbedArgs a, const EmbedMeta* meta) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rows = a.B * a.S_cap;
    for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
        const EmbedMeta m = meta[row];
        float* out = a.hidden + (size_t)row * a.d;
        for (int c = lane * 4; c < a.d; c += 256) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (m.tok >= 0) {
                const uint2 t = *(const uint2*)(a.tok_emb + (size_t)m.tok * a.d + c);
                v[0] = bf16lo(t.x); v[1] = bf16hi(t.x); v[2] = bf16lo(t.y); v[3] = bf16hi(t.y);
            }
            if (m.patch >= 0) {
                const float4 p = *(const float4*)(a.patch_emb + (size_t)m.patch * a.d + c);
                v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
            }
            if (m.c[0] >= 0) {
                const uint2 e0 = *(const uint2*)(a.x_emb + (size_t)m.c[0] * a.d + c);
                const uint2 e1 = *(const uint2*)(a.y_emb + (size_t)m.c[1] * a.d + c);
                const uint2 e2 = *(const uint2*)(a.x_emb + (size_t)m.c[2] * a.d + c);
                const uint2 e3 = *(const uint2*)(a.y_emb + (size_t)m.c[3] * a.d + c);
                v[0] += ((bf16lo(e0.x) + bf16lo(e1.x)) + bf16lo(e2.x)) + bf16lo(e3.x);
                v[1] += ((bf16hi(e0.x) + bf16hi(e1.x)) + bf16hi(e2.x)) + bf16hi(e3.x);
                v[2] += ((bf16lo(e0.y) + bf16lo(e1.y)) + bf16lo(e2.y)) + bf16lo(e3.y);
                v[3] += ((bf16hi(e0.y) + bf16hi(e1.y)) + bf16hi(e2.y)) + bf16hi(e3.y);
            }
            if (a.hidden_tiled) *(float4*)(a.hidden + ht_off(row, c, a.d)) = make_float4(v[0], v[1], v[2], v[3]);
            else *(float4*)(out + c) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

size_t embed_meta_bytes(int B, int S_cap) { return (size_t)B * S_cap * sizeof(EmbedMeta); }

void embed_assemble(const EmbedArgs& a, void* meta_ws, mgStream_t stream) {
    EmbedMeta* meta = (EmbedMeta*)meta_ws;
    const size_t sh = ((a.P + 15) & ~15) + 256 * sizeof(int);
    MG_LAUNCH(embed_index_kernel, dim3(a.B), dim3(256), sh, stream, a, meta);
    int blocks = (a.B * a.S_cap + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    MG_LAUNCH(embed_gather_kernel, dim3(blocks), dim3(256), 0, stream, a, (const EmbedMeta*)meta);
}

// decoder token embedding: h[row] = shared[ids[row]]  (stock:1140)
__global__ __launch_bounds__(256) void embed_rows_kernel(const int64_t* ids, const uint16_t* tok_emb, float* h, int rows, int d, int V,
                                                    int* err, uint16_t* x_pk, int x_ld, int x_col0) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
        long long id = ids[row];
        if (id < 0 || id >= V) { if (lane == 0) atomicAdd(err, 1); id = 0; }
        for (int c = lane * 4; c < d; c += 256) {
            const uint2 t = *(const uint2*)(tok_emb + (size_t)id * d + c);
            *(float4*)(h + (size_t)row * d + c) = make_float4(bf16lo(t.x), bf16hi(t.x), bf16lo(t.y), bf16hi(t.y));
            if (x_pk) *(uint2*)(x_pk + pk_off(row, x_col0 + c, x_ld)) = t;      // 4 consecutive k of one packed row
        }
    }
}
void embed_rows(const int64_t* ids, const uint16_t* tok_emb, float* h, int rows, int d, int V, int* err, mgStream_t stream,
                uint16_t* x_pk, int x_ld, int x_col0) {
    int blocks = (rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    MG_LAUNCH(embed_rows_kernel, dim3(blocks), dim3(256), 0, stream, ids, tok_emb, h, rows, d, V, err, x_pk, x_ld, x_col0);
}

// Decode step, first launch: token embedding (stock:1140) + the first decoder layer's RMSNorm (stock:611-616) in one
// pass.  Same arithmetic and summation order as embed_rows followed by rmsnorm_pack (one wave per row, 8 features per
// lane and step).  Writes h (fp32), x_pk = bf16(RMSNorm(h) * gain) and x2_pk = the embedding row itself (packed window).
__global__ __launch_bounds__(256) void embed_norm_rows_kernel(const int64_t* ids, const uint16_t* tok_emb, float* h, const float* gain,
                                                         uint16_t* x_pk, uint16_t* x2_pk, int x2_ld, int x2_col0, int rows, int d, int V,
                                                         int* err, float eps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nch = d >> 3;
    for (int m = blockIdx.x * 4 + w; m < rows; m += gridDim.x * 4) {
        long long id = ids[m];
        if (id < 0 || id >= V) { if (lane == 0) atomicAdd(err, 1); id = 0; }
        const uint16_t* src = tok_emb + (size_t)id * d;
        float ss = 0.f;
        for (int c = lane; c < nch; c += 64) {
            const uint4 t = ld16(src + c * 8);
            const float a0 = bf16lo(t.x), a1 = bf16hi(t.x), a2 = bf16lo(t.y), a3 = bf16hi(t.y);
            const float b0 = bf16lo(t.z), b1 = bf16hi(t.z), b2 = bf16lo(t.w), b3 = bf16hi(t.w);
            ss += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3 + b0 * b0 + b1 * b1 + b2 * b2 + b3 * b3;
        }
        ss = wave_sum(ss);
        const float r = rsqrtf(ss / (float)d + eps);
        for (int c = lane; c < nch; c += 64) {
            const uint4 t = ld16(src + c * 8);
            const float a[8] = {bf16lo(t.x), bf16hi(t.x), bf16lo(t.y), bf16hi(t.y), bf16lo(t.z), bf16hi(t.z), bf16lo(t.w), bf16hi(t.w)};
            const float4 g0 = *(const float4*)(gain + c * 8), g1 = *(const float4*)(gain + c * 8 + 4);
            *(float4*)(h + (size_t)m * d + c * 8) = make_float4(a[0], a[1], a[2], a[3]);
            *(float4*)(h + (size_t)m * d + c * 8 + 4) = make_float4(a[4], a[5], a[6], a[7]);
            st16(x_pk + pk_off(m, c * 8, d),
                 make_uint4(pack_bf16(g0.x * (a[0] * r), g0.y * (a[1] * r)), pack_bf16(g0.z * (a[2] * r), g0.w * (a[3] * r)),
                            pack_bf16(g1.x * (a[4] * r), g1.y * (a[5] * r)), pack_bf16(g1.z * (a[6] * r), g1.w * (a[7] * r))));
            if (x2_pk) st16(x2_pk + pk_off(m, x2_col0 + c * 8, x2_ld), t);
        }
    }
}
void embed_norm_rows(const int64_t* ids, const uint16_t* tok_emb, float* h, const float* gain, uint16_t* x_pk, uint16_t* x2_pk, int x2_ld,
                     int x2_col0, int rows, int d, int V, int* err, float eps, mgStream_t stream) {
    int blocks = (rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    MG_LAUNCH(embed_norm_rows_kernel, dim3(blocks), dim3(256), 0, stream, ids, tok_emb, h, gain, x_pk, x2_pk, x2_ld, x2_col0, rows, d, V, err, eps);
}

}  // namespace mg
