// Launchers of k_ocr.hip (ChemicalOCR stage).
#pragma once
#include "mg_kernels.h"

namespace mg {
void ocr_layernorm_pack(float* h, const float* w, const float* b, const float* add_bias, uint16_t* x_pk, float* out_f32, int M, int d,
                        int Kaug, float eps, mgStream_t st);
void ocr_gelu_pack(const float* in, uint16_t* y_pk, int M, int N, int Kaug, mgStream_t st);
void ocr_silu_mul_pack(const float* in, uint16_t* y_pk, int M, int I, mgStream_t st);
void ocr_add_pos(const float* patch, const uint16_t* pos, const int* pos_ids, const uint8_t* patch_mask, uint8_t* vmask, float* hidden, int N, int P,
                 int P_cap, int d, mgStream_t st);
void ocr_pixel_shuffle_pack(const float* vis, uint16_t* x_pk, int N, int g, int P_cap, int e, int sf, mgStream_t st);
void ocr_merge_embed(const int64_t* ids, const uint16_t* tok_emb, const float* feats, float* h, int B, int L, int T_cap, int d, int V,
                     int image_token, int per_seq, int* err, mgStream_t st);
void ocr_rope_heads(const float* qkv, int B, int T, int T_cap, int H, int KV, float theta, uint16_t* Q, uint16_t* K, uint16_t* Vt,
                    uint16_t* Kc, uint16_t* Vc, int cap, mgStream_t st);
void ocr_rope_table(float* cs, int positions, float theta, mgStream_t st);
void ocr_silu_mul_rows(const float* in, const RowScale& rs, uint16_t* y_pk, int M, int I, mgStream_t st);
void ocr_pack_aug(const float* W, const float* bias, float scale, uint16_t* dst, int row0, int N, int K, int Kaug, int Nfill, int rstride, mgStream_t st);
void ocr_init(int64_t* out_ids, int* unfinished, int* counters, int rows, int max_new, int64_t pad, mgStream_t st);
void ocr_slots_init(int* unfinished, int* pos, int* img, int* pool, int64_t* next_ids, int slots, int* ctr, int N, mgStream_t st);
void ocr_fill_ints(int* p, int v, int n, mgStream_t st);
void ocr_add_int(int* dst, const int* src, mgStream_t st);
void ocr_set_int(int* dst, int v, mgStream_t st);
// row-major [M][d] fp32 <-> tiled (ht_off) copies, M a multiple of 32
void ocr_tile_f32(const float* src, float* dst, int M, int d, int to_tiled, mgStream_t st);
// lens (nullable): prompts of different lengths, left-aligned in their rows: row b's last position is lens[b] - 1 and its keys are [0, lens[b])
void ocr_row_maps(int* last_rows, int* all_rows, uint8_t* key_mask, int B, int T, int T_cap, mgStream_t st, const int* lens = nullptr);
void ocr_len_delta(const int* lens, int* delta, int N, int L, int* err, mgStream_t st);     // delta[n] = lens[n] - L (lengths outside [1, L] counted in *err)
// engine.hip: sets the thread-local message mg_last_error() returns
int fail_msg(int code, const char* msg);
}  // namespace mg
