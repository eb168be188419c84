// Layout / normalisation kernels: weight packing into MFMA fragment tiles, patch im2col, fused RMSNorm + pack.
// All are HBM-bound streaming kernels: 16-byte accesses per lane, grid-stride.
#include "mg_kernels.h"

namespace mg {

MG_DEV float load_elem(const void* src, int is_bf16, size_t i) {
    return is_bf16 ? bf16_to_f32(((const uint16_t*)src)[i]) : ((const float*)src)[i];
}

// HF nn.Linear weight [N][K] (row-major) -> packed fragment tiles [Npad/32][K/16]; rows >= N are zero.
__global__ __launch_bounds__(256) void pack_weight_kernel(const void* src, int is_bf16, int N, int K, uint16_t* dst, int Npad) {
    const size_t nchunk = (size_t)(Npad >> 5) * (size_t)(K >> 4) * 64;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += (size_t)gridDim.x * blockDim.x) {
        const size_t tile = c >> 6;
        const int l = (int)(c & 63);
        const int rt = (int)(tile / (size_t)(K >> 4)), kt = (int)(tile % (size_t)(K >> 4));
        const int row = rt * 32 + (l & 31), k = kt * 16 + 8 * (l >> 5);
        uint32_t w[4] = {0, 0, 0, 0};
        if (row < N) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w[j] = pack_bf16(load_elem(src, is_bf16, (size_t)row * K + k + 2 * j),
                                 load_elem(src, is_bf16, (size_t)row * K + k + 2 * j + 1));
        }
        st16(dst + c * 8, make_uint4(w[0], w[1], w[2], w[3]));
    }
}
void pack_weight(const void* src, int src_is_bf16, int N, int K, uint16_t* dst, int Npad, mgStream_t stream) {
    const size_t nchunk = (size_t)(Npad >> 5) * (size_t)(K >> 4) * 64;
    int blocks = (int)((nchunk + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    MG_LAUNCH(pack_weight_kernel, dim3(blocks), dim3(256), 0, stream, src, src_is_bf16, N, K, dst, Npad);
}

__global__ __launch_bounds__(256) void convert_f32_kernel(const void* src, int is_bf16, float* dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = load_elem(src, is_bf16, i);
}
__global__ __launch_bounds__(256) void convert_bf16_kernel(const void* src, int is_bf16, uint16_t* dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = is_bf16 ? ((const uint16_t*)src)[i] : f32_to_bf16(((const float*)src)[i]);
}
static int grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b ? b : 1));
}
void convert_to_f32(const void* src, int src_is_bf16, float* dst, size_t n, mgStream_t stream) {
    MG_LAUNCH(convert_f32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, src, src_is_bf16, dst, n);
}
void convert_to_bf16(const void* src, int src_is_bf16, uint16_t* dst, size_t n, mgStream_t stream) {
    MG_LAUNCH(convert_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, stream, src, src_is_bf16, dst, n);
}

// Patch embedding as a GEMM (stock:254-280: Conv2d(k = s = patch) == [B*P, C*ps*ps] x [C*ps*ps, d]):
// X[m = b*P + py*n + px][k = c*ps*ps + ky*ps + kx] = pix[b][c][py*ps + ky][px*ps + kx], written packed.
__global__ __launch_bounds__(256) void im2col_pack_kernel(const float* pix, uint16_t* x_pk, int B, int C, int I, int ps) {
    const int n = I / ps, P = n * n, K = C * ps * ps, M = B * P;
    const size_t nchunk = (size_t)((M + 31) >> 5) * (size_t)(K >> 4) * 64;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += (size_t)gridDim.x * blockDim.x) {
        const size_t tile = c >> 6;
        const int l = (int)(c & 63);
        const int rt = (int)(tile / (size_t)(K >> 4)), kt = (int)(tile % (size_t)(K >> 4));
        const int m = rt * 32 + (l & 31), k = kt * 16 + 8 * (l >> 5);
        uint32_t w[4] = {0, 0, 0, 0};
        if (m < M) {
            const int b = m / P, p = m - b * P, py = p / n, px = p - py * n;
            const int ch = k / (ps * ps), kk = k - ch * ps * ps, ky = kk / ps, kx = kk - ky * ps;
            const float* s = pix + (((size_t)b * C + ch) * I + (size_t)(py * ps + ky)) * I + (size_t)(px * ps + kx);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = pack_bf16(s[2 * j], s[2 * j + 1]);
        }
        st16(x_pk + c * 8, make_uint4(w[0], w[1], w[2], w[3]));
    }
}
void im2col_pack(const float* pix, uint16_t* x_pk, int B, int C, int I, int ps, mgStream_t stream) {
    const int n = I / ps, M = B * n * n, K = C * ps * ps;
    const size_t nchunk = (size_t)((M + 31) >> 5) * (size_t)(K >> 4) * 64;
    MG_LAUNCH(im2col_pack_kernel, dim3(grid_for(nchunk)), dim3(256), 0, stream, pix, x_pk, B, C, I, ps);
}

__global__ __launch_bounds__(256) void pack_e1_kernel(const float* e1, int B, int M, int M_pad, int d, uint16_t* e1_pk, int* row_map,
                                                 const uint8_t* enc_mask, int S_cap, uint8_t* xmask) {
    const int nch = d >> 3;
    const size_t total = (size_t)B * M_pad * nch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nch);
        const size_t row = i / nch;
        const int b = (int)(row / M_pad), j = (int)(row - (size_t)b * M_pad);
        uint4 o = make_uint4(0, 0, 0, 0);
        if (j < M) {
            const float* p = e1 + ((size_t)b * M + j) * d + c * 8;
            const float4 x = *(const float4*)p, y = *(const float4*)(p + 4);
            o = make_uint4(pack_bf16(x.x, x.y), pack_bf16(x.z, x.w), pack_bf16(y.x, y.y), pack_bf16(y.z, y.w));
        }
        st16(e1_pk + pk_off((int)row, c * 8, d), o);
        if (c == 0) row_map[row] = j < M ? j : -1;
    }
    if (xmask) {
        const size_t nx = (size_t)B * (M_pad + S_cap);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nx; i += (size_t)gridDim.x * blockDim.x) {
            const int b = (int)(i / (M_pad + S_cap)), j = (int)(i - (size_t)b * (M_pad + S_cap));
            xmask[i] = j < M ? 1 : (j < M_pad ? 0 : enc_mask[(size_t)b * S_cap + (j - M_pad)]);
        }
    }
}
void pack_e1(const float* e1, int B, int M, int M_pad, int d, uint16_t* e1_pk, int* row_map, const uint8_t* enc_mask, int S_cap,
             uint8_t* xmask, mgStream_t stream) {
    MG_LAUNCH(pack_e1_kernel, dim3(grid_for((size_t)B * (M_pad + S_cap) * (d >> 3))), dim3(256), 0, stream, e1, B, M, M_pad, d, e1_pk,
              row_map, enc_mask, S_cap, xmask);
}

// Fused RMSNorm + bf16 pack (stock:293-306: fp32 variance, x*rsqrt(var+eps), then *gain; `scale` folds the
// d_model^-0.5 of the tied lm_head, stock:1554-1555).  One wave per row, 8 consecutive features per lane
// per step -> one 16-byte chunk of the packed operand.  Rows >= M of the last 32-row tile are zero-filled by
// the caller's buffer initialisation (their GEMM results are discarded anyway).
__global__ __launch_bounds__(256) void rmsnorm_pack_kernel(const float* h, const float* gain, uint16_t* x_pk, float* out_f32,
                                                      const int* dst_row, int M, int d, float eps, float scale) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nch = d >> 3;
    for (int m = blockIdx.x * 4 + w; m < M; m += gridDim.x * 4) {
        const float* row = h + (size_t)m * d;
        float ss = 0.f;
        for (int c = lane; c < nch; c += 64) {
            const float4 a = *(const float4*)(row + c * 8), b = *(const float4*)(row + c * 8 + 4);
            ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
        }
        ss = wave_sum(ss);
        const float r = rsqrtf(ss / (float)d + eps);
        for (int c = lane; c < nch; c += 64) {
            const float4 a = *(const float4*)(row + c * 8), b = *(const float4*)(row + c * 8 + 4);
            const float4 g0 = *(const float4*)(gain + c * 8), g1 = *(const float4*)(gain + c * 8 + 4);
            float v[8] = {g0.x * (a.x * r), g0.y * (a.y * r), g0.z * (a.z * r), g0.w * (a.w * r),
                          g1.x * (b.x * r), g1.y * (b.y * r), g1.z * (b.z * r), g1.w * (b.w * r)};
            if (scale != 1.0f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= scale;
            }
            if (out_f32) {
                *(float4*)(out_f32 + (size_t)m * d + c * 8) = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)(out_f32 + (size_t)m * d + c * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            const int mo = dst_row ? dst_row[m] : m;
            if (x_pk && mo >= 0)
                st16(x_pk + pk_off(mo, c * 8, d),
                     make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])));
        }
    }
}
void rmsnorm_pack(const float* h, const float* gain, uint16_t* x_pk, float* out_f32, int M, int d, float eps,
                  float scale, mgStream_t stream) {
    int blocks = (M + 3) / 4;
    if (blocks > 65536) blocks = 65536;     // one row per wave up to 256 K rows
    if (blocks < 1) blocks = 1;
    MG_LAUNCH(rmsnorm_pack_kernel, dim3(blocks), dim3(256), 0, stream, h, gain, x_pk, out_f32, (const int*)nullptr, M, d, eps, scale);
}
// RMSNorm + pack from the tiled fp32 layout: one workgroup per 32-row tile; thread = (row, one of 8 feature slices), so the
// 32 lanes of a half-wave read 512 contiguous bytes per feature group and write 512 contiguous bytes of a packed tile.
__global__ __launch_bounds__(256) void rmsnorm_pack_tiled_kernel(const float* h, const float* gain, uint16_t* x_pk, float* out_f32, int M, int d,
                                                            float eps) {
    MG_DYN_SMEM(smem);
    float* red = (float*)smem;                       // [8][32]
    const int tid = threadIdx.x, r = tid & 31, p = tid >> 5;
    const int nch = d >> 3;                          // 8-feature chunks per row
    for (int rt = blockIdx.x; rt < (M >> 5); rt += gridDim.x) {
        const int m = rt * 32 + r;
        float ss = 0.f;
        for (int c = p; c < nch; c += 8) {
            const float4 a = *(const float4*)(h + ht_off(m, c * 8, d)), b = *(const float4*)(h + ht_off(m, c * 8 + 4, d));
            ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
        }
        red[p * 32 + r] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) tot += red[i * 32 + r];
        const float rr = rsqrtf(tot / (float)d + eps);
        for (int c = p; c < nch; c += 8) {
            const float4 a = *(const float4*)(h + ht_off(m, c * 8, d)), b = *(const float4*)(h + ht_off(m, c * 8 + 4, d));
            const float4 g0 = *(const float4*)(gain + c * 8), g1 = *(const float4*)(gain + c * 8 + 4);
            const float v[8] = {g0.x * (a.x * rr), g0.y * (a.y * rr), g0.z * (a.z * rr), g0.w * (a.w * rr),
                                g1.x * (b.x * rr), g1.y * (b.y * rr), g1.z * (b.z * rr), g1.w * (b.w * rr)};
            if (out_f32) {
                *(float4*)(out_f32 + (size_t)m * d + c * 8) = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)(out_f32 + (size_t)m * d + c * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (x_pk)
                st16(x_pk + pk_off(m, c * 8, d),
                     make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])));
        }
        __syncthreads();
    }
}
void rmsnorm_pack_tiled(const float* h_tiled, const float* gain, uint16_t* x_pk, float* out_f32, int M, int d, float eps,
                        mgStream_t stream) {
    int blocks = M >> 5;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    MG_LAUNCH(rmsnorm_pack_tiled_kernel, dim3(blocks), dim3(256), 8 * 32 * sizeof(float), stream, h_tiled, gain, x_pk, out_f32, M, d, eps);
}
void rmsnorm_pack_rows(const float* h, const float* gain, uint16_t* x_pk, const int* dst_row, int M, int d, float eps,
                       float scale, mgStream_t stream) {
    int blocks = (M + 3) / 4;
    if (blocks > 65536) blocks = 65536;     // one row per wave up to 256 K rows
    if (blocks < 1) blocks = 1;
    MG_LAUNCH(rmsnorm_pack_kernel, dim3(blocks), dim3(256), 0, stream, h, gain, x_pk, (float*)nullptr, dst_row, M, d, eps, scale);
}

// Decode-step fusion: residual add of a split-K projection (KS partial slabs, summed in slab order) + the next
// sub-layer's RMSNorm + bf16 pack.  One workgroup per sequence row, 4 features per thread; all 1+KS loads of a thread
// are issued together (one L2 round trip), the row stays in registers between the two passes.
template <int KSMAX>
__global__ __launch_bounds__(256) void add_norm_pack_kernel(float* h, Slabs add, const float* gain, uint16_t* x_pk, int M, int d, float eps,
                                                       float scale) {
    MG_DYN_SMEM(smem);
    float* red = (float*)smem;
    const int tid = threadIdx.x, m = blockIdx.x;
    const int nq = d >> 2;                       // float4 groups per row
    float* row = h + (size_t)m * d;
    float ss = 0.f;
    float4 v[4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
        const int c = tid + 256 * ci;
        v[ci] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nq) {
            float4 part[KSMAX];
            const float4 a = *(const float4*)(row + c * 4);
#pragma unroll
            for (int s = 0; s < KSMAX; ++s)
                if (s < add.KS) part[s] = *(const float4*)(add.P + (size_t)s * add.stride + (size_t)m * add.ldp + c * 4);
            float4 t = a;
#pragma unroll
            for (int s = 0; s < KSMAX; ++s)
                if (s < add.KS) { t.x += part[s].x; t.y += part[s].y; t.z += part[s].z; t.w += part[s].w; }
            v[ci] = t;
            *(float4*)(row + c * 4) = t;
            ss += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
        }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    ss = (red[0] + red[1]) + (red[2] + red[3]);
    const float r = rsqrtf(ss / (float)d + eps);
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
        const int c = tid + 256 * ci;
        if (c < nq) {
            const float4 g = *(const float4*)(gain + c * 4);
            float o0 = g.x * (v[ci].x * r), o1 = g.y * (v[ci].y * r), o2 = g.z * (v[ci].z * r), o3 = g.w * (v[ci].w * r);
            if (scale != 1.0f) { o0 *= scale; o1 *= scale; o2 *= scale; o3 *= scale; }
            *(uint2*)(x_pk + pk_off(m, c * 4, d)) = make_uint2(pack_bf16(o0, o1), pack_bf16(o2, o3));
        }
    }
}
void add_norm_pack(float* h, const Slabs& add, const float* gain, uint16_t* x_pk, int M, int d, float eps, float scale,
                   mgStream_t stream) {
    if (add.KS <= 4) MG_LAUNCH((add_norm_pack_kernel<4>), dim3(M), dim3(256), 64, stream, h, add, gain, x_pk, M, d, eps, scale);
    else if (add.KS <= 8) MG_LAUNCH((add_norm_pack_kernel<8>), dim3(M), dim3(256), 64, stream, h, add, gain, x_pk, M, d, eps, scale);
    else MG_LAUNCH((add_norm_pack_kernel<16>), dim3(M), dim3(256), 64, stream, h, add, gain, x_pk, M, d, eps, scale);
}

// y_pk = bf16(relu(sum of the split-K slabs)), packed: the FFN activation between wi and wo (stock:318-321)
__global__ __launch_bounds__(256) void relu_pack_kernel(Slabs in, uint16_t* y_pk, int M, int N) {
    const int nch = N >> 3;
    const size_t total = (size_t)M * nch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / nch), c = (int)(i - (size_t)m * nch);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < in.KS; ++s) {
            const float* p = in.P + (size_t)s * in.stride + (size_t)m * in.ldp + c * 8;
            const float4 pa = *(const float4*)p, pb = *(const float4*)(p + 4);
            v[0] += pa.x; v[1] += pa.y; v[2] += pa.z; v[3] += pa.w; v[4] += pb.x; v[5] += pb.y; v[6] += pb.z; v[7] += pb.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        st16(y_pk + pk_off(m, c * 8, N),
             make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])));
    }
}
void relu_pack(const Slabs& in, uint16_t* y_pk, int M, int N, mgStream_t stream) {
    const size_t total = (size_t)M * (N >> 3);
    MG_LAUNCH(relu_pack_kernel, dim3(grid_for(total)), dim3(256), 0, stream, in, y_pk, M, N);
}

}  // namespace mg
