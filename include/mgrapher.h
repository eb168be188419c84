/* mgrapher.h — C ABI of libmgrapher_hip.so: the MI355X-native MarkushGrapher-2 VTL encoder + CXSMILES decoder
 * forward path.  (Work in progress: engine-level entry points are added below as they land.)
 */
#ifndef MGRAPHER_H
#define MGRAPHER_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { MG_OK = 0, MG_E_SHAPE = -1, MG_E_ARG = -2, MG_E_STATE = -3, MG_E_HIP = -4, MG_E_UNSUPPORTED = -5,
       MG_E_WORKSPACE = -6, MG_E_KEY = -7 };

int mgk_pack_weight(void* stream, const void* src, int src_is_bf16, int N, int K, void* dst_pk, int Npad);
int mgk_rmsnorm_pack(void* stream, const float* h, const float* gain, void* x_pk, float* out_f32, int M, int d,
                     float eps, float scale);
int mgk_im2col_pack(void* stream, const float* pix, void* x_pk, int B, int C, int I, int ps);
int mgk_gemm(void* stream, int mode, int epi, const void* X_pk, const void* W_pk, int M, int N, int K, float* out_f32,
             int ldo, const float* bias, void* out_pk);
int mgk_gemm_heads(void* stream, int mode, const void* X_pk, const void* W_pk, int M, int N, int K, void* p0, void* p1,
                   void* p2, int f0, int f1, int f2, int H, int S_in, int S_cap, const int* row_map, int pos);

#ifdef __cplusplus
}
#endif
#endif
