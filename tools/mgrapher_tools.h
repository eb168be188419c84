/* Instrumented entry points that exist ONLY in the tools build of the library (`python markushgrapher_amd/csrc/build.py tools`
 * -> tools/_build/libmgrapher_tools.so, compiled with -DMG_TOOLS).  The product library libmgrapher_hip.so does not contain them,
 * nor the what-if GEMM variants behind MG_GEMM_EXP (which compute WRONG results on purpose, for timing only). */
#pragma once
#include "../include/mgrapher.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Phase-stamped copy of the decoder cross-attention kernel (tools/trace_attn.py): trace[(workgroup*8 + wave)*8 + k]. */
int mgk_attention_step_trace(void* stream, const void* q, const void* Kc, const void* Vc, void* ctx_pk, int rows, int H, int cap,
                             const int* len, long long* trace);
/* Phase-stamped copy of the FFN-wo residual projection (32 rows, 16 waves): trace[(workgroup*16 + wave)*8 + k] = shader
 * clock at phase k (tools/trace_resid.py). */
int mgk_gemm_resid_trace(void* stream, const void* X_pk, const void* W_pk, float* h, const float* gain, void* x_pk, float* part, int N,
                         int K, const float* rs_part, long long* trace);
#ifdef __cplusplus
}
#endif
