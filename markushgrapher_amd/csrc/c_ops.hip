// C-ABI operator entry points (kernel-level): what the parity tests call to compare each HIP kernel with the
// oracle.  Plain pointers and sizes only; all pointers are device pointers; work is enqueued on `stream`.
#include "mg_kernels.h"
#include "../../include/mgrapher.h"

using namespace mg;

extern "C" {

int mgk_pack_weight(void* stream, const void* src, int src_is_bf16, int N, int K, void* dst_pk, int Npad) {
    if ((K & 15) || (Npad & 31) || Npad < N) return MG_E_SHAPE;
    pack_weight(src, src_is_bf16, N, K, (uint16_t*)dst_pk, Npad, (mgStream_t)stream);
    return MG_OK;
}

int mgk_rmsnorm_pack(void* stream, const float* h, const float* gain, void* x_pk, float* out_f32, int M, int d,
                     float eps, float scale) {
    if (d & 15) return MG_E_SHAPE;
    rmsnorm_pack(h, gain, (uint16_t*)x_pk, out_f32, M, d, eps, scale, (mgStream_t)stream);
    return MG_OK;
}

int mgk_im2col_pack(void* stream, const float* pix, void* x_pk, int B, int C, int I, int ps) {
    if ((ps & 7) || (I % ps)) return MG_E_SHAPE;
    im2col_pack(pix, (uint16_t*)x_pk, B, C, I, ps, (mgStream_t)stream);
    return MG_OK;
}

// mode: 0 = tiled (large M), 1 = row-streaming (decode step)
int mgk_gemm(void* stream, int mode, int epi, const void* X_pk, const void* W_pk, int M, int N, int K, float* out_f32,
             int ldo, const float* bias, void* out_pk) {
    if ((K & 63) || epi < 0 || epi > EPI_PK) return MG_E_SHAPE;
    GemmArgs a{};
    a.X = (const uint16_t*)X_pk; a.W = (const uint16_t*)W_pk; a.M = M; a.N = N; a.K = K;
    a.out_f32 = out_f32; a.ldo = ldo; a.bias = bias; a.out_pk = (uint16_t*)out_pk;
    if (mode == 0) gemm(a, epi, (mgStream_t)stream); else gemm_rows(a, epi, (mgStream_t)stream);
    return MG_OK;
}

int mgk_gemm_heads(void* stream, int mode, const void* X_pk, const void* W_pk, int M, int N, int K, void* p0, void* p1,
                   void* p2, int f0, int f1, int f2, int H, int S_in, int S_cap, const int* row_map, int pos) {
    if ((K & 63) || (N % (H * 64))) return MG_E_SHAPE;
    GemmArgs a{};
    a.X = (const uint16_t*)X_pk; a.W = (const uint16_t*)W_pk; a.M = M; a.N = N; a.K = K;
    a.heads.ptr[0] = (uint16_t*)p0; a.heads.ptr[1] = (uint16_t*)p1; a.heads.ptr[2] = (uint16_t*)p2;
    a.heads.fmt[0] = f0; a.heads.fmt[1] = f1; a.heads.fmt[2] = f2;
    a.heads.inner = H * 64; a.heads.H = H; a.heads.S_in = S_in; a.heads.S_cap = S_cap; a.heads.row_map = row_map;
    a.heads.pos = pos;
    if (mode == 0) gemm(a, EPI_HEADS, (mgStream_t)stream); else gemm_rows(a, EPI_HEADS, (mgStream_t)stream);
    return MG_OK;
}

}  // extern "C"
