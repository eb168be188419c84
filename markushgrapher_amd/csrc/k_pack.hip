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
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    MG_LAUNCH(rmsnorm_pack_kernel, dim3(blocks), dim3(256), 0, stream, h, gain, x_pk, out_f32, (const int*)nullptr, M, d, eps, scale);
}
void rmsnorm_pack_rows(const float* h, const float* gain, uint16_t* x_pk, const int* dst_row, int M, int d, float eps,
                       float scale, mgStream_t stream) {
    int blocks = (M + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    MG_LAUNCH(rmsnorm_pack_kernel, dim3(blocks), dim3(256), 0, stream, h, gain, x_pk, (float*)nullptr, dst_row, M, d, eps, scale);
}

}  // namespace mg
