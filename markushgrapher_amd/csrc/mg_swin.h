// Launchers of k_swin.hip: the OCSR vision branch "e1"  - a Swin encoder as stock
// transformers models/swin/modeling_swin.py states it (the importable upstream of MolScribe's timm swin_base_patch4_window12_384).
#pragma once
#include "mg_kernels.h"

namespace mg {

// dst [B][C][I][I] = bilinear resize (align_corners = false, no antialias: torch.nn.functional.interpolate) of src [B][C][S][S],
// then x * scale[c] + shift[c].  C <= 4.
struct SwinPixAffine { float scale[4], shift[4]; };
void swin_resize(const float* src, float* dst, int B, int C, int S, int I, const SwinPixAffine& af, mgStream_t st);
// pixels [B][C][I][I] fp32 -> packed bf16 im2col matrix [B*g*g][Kp], k = (c*ps + dy)*ps + dx, columns >= C*ps*ps zero
// (Conv2d with kernel = stride = patch as a matrix product, stock:254-286)
void swin_im2col_pack(const float* pix, uint16_t* x_pk, int B, int C, int I, int ps, int Kp, mgStream_t st);

// LayerNorm over the C features of M rows of the fp32 residual stream (stock nn.LayerNorm: biased variance, eps inside the root):
//   h_in      the rows: TILED (ht_off, in_tiled = 1: what the GEMMs' residual epilogue EPI_RESID_NORM reads and writes) or row-major
//             (in_tiled = 0: the output of a plain fp32-store GEMM - patch embedding, patch-merging reduction)
//   x_pk (nullable)    packed bf16 [M][C] = LN(row) * w + b
//   out_f32 (nullable) row-major fp32 [M][C] of the same
//   h_out (nullable)   TILED [M][C]: the row itself + add_bias (nullable): the residual stream the following projections accumulate into,
//                      with the bias of the projection (o_proj / fc2) whose product is added later in the same sub-layer (the residual
//                      epilogue has no bias slot); may be h_in itself when that is tiled
// merge_R > 0: patch merging (stock:309-326) - h_in is the TILED merge_R^2 map of width C / 4, output row (b, i, j) of the
//   (merge_R/2)^2 map normalises the concatenation [h(2i, 2j) | h(2i+1, 2j) | h(2i, 2j+1) | h(2i+1, 2j+1)]
struct SwinLnArgs {
    const float* h_in;
    int in_tiled;
    float* h_out;
    int h_out_norm;        // 1: h_out receives the NORMALISED row (embeddings.norm: its output is the residual stream)
    const float* w;
    const float* b;
    const float* add_bias;
    uint16_t* x_pk;
    float* out_f32;
    int M, C;
    int merge_R;
    float eps;
    int kaug;              // 0 = C; > C: x_pk has kaug columns per row, columns C .. kaug - 1 = [1, 0, ...] (bias-in-K projections of the OCR tower)
};
bool swin_ln_supported(int C);      // widths of the Swin branch (powers of two 64 .. 4096); the kernel also takes 768
void swin_layernorm(const SwinLnArgs& a, mgStream_t st);

// (Shifted-)window attention of one Swin block (stock:418-468, 529-563, 584-626), head dim 32:
//   qkv   packed bf16 [M][3C] = [q | k | v] of the NATURAL token order m = (b, y, x) of the R x R map
//   ctx   packed bf16 [M][C], natural order
//   table relative-position biases TRANSPOSED to [H][(2w-1)^2] fp32
// A workgroup takes one window of one image and up to 4 heads (one wave each); the window's rows are gathered through the cyclic
// shift (torch.roll by -shift) and scattered back through its inverse; scores = q k^T * 32^-0.5 + table[dy, dx] + (-100 where the
// two tokens lie in different regions of the shifted map), softmax in fp32, weights rounded to bf16 for the second product and
// normalised by the sum of the rounded weights.  w * w must be a multiple of 16 (w = 4, 8, 12), R a multiple of w.
struct SwinAttnArgs {
    const uint16_t* qkv;
    uint16_t* ctx;
    const float* table;
    int B, R, C, H, w, shift;
};
bool swin_attention_supported(int w, int R, int C, int H);
void swin_attention(const SwinAttnArgs& a, mgStream_t st);

// table [n][H] -> [H][n]
void swin_transpose_f32(const float* src, float* dst, int n, int H, mgStream_t st);

// engine.hip: sets the thread-local message mg_last_error() returns
int fail_msg(int code, const char* msg);
}  // namespace mg
