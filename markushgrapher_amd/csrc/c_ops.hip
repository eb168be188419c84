// C-ABI operator entry points (kernel-level): what the parity tests call to compare each HIP kernel with the
// oracle.  Plain pointers and sizes only; all pointers are device pointers; work is enqueued on `stream`.
#include "mg_kernels.h"
#include <stdlib.h>
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
int mgk_set_rows_split(int mode) { gemm_rows_set_split(mode); return MG_OK; }
int mgk_set_resid_f16(int on) { gemm_rows_set_resid_f16(on); return MG_OK; }
int mgk_set_rows_ft2(int mode) { gemm_rows_set_ft2(mode); return MG_OK; }
int mgk_gemm(void* stream, int mode, int epi, const void* X_pk, const void* W_pk, int M, int N, int K, float* out_f32,
             int ldo, const float* bias, void* out_pk) {
    if ((K & 63) || epi < 0 || (epi > EPI_PK && epi != EPI_PK_GELU)) return MG_E_SHAPE;
    if (epi == EPI_PK_GELU && !(mode == 0 && gemm_has_gelu_epilogue(M, N))) return MG_E_UNSUPPORTED;
    GemmArgs a{};
    a.X = (const uint16_t*)X_pk; a.W = (const uint16_t*)W_pk; a.M = M; a.N = N; a.K = K;
    a.out_f32 = out_f32; a.ldo = ldo; a.bias = bias; a.out_pk = (uint16_t*)out_pk;
    if (mode == 0) gemm(a, epi, (mgStream_t)stream); else gemm_rows(a, epi, (mgStream_t)stream);
    return MG_OK;
}

// Deferred-RMSNorm pair of the encoder (test entry):
//   epi = EPI_RESID_NORM (5): h_tiled (fp32, layout ht_off) += X W^T, x_out_pk = pack(bf16(h * gain)), part[M][part_ld] partial sums
//   epi = EPI_PK / EPI_PK_RELU (3 / 2): out_pk = pack(bf16(relu?(X W^T * r(m)))) with r from rs_part[M][rs_nparts] (null: r = 1)
int mgk_gemm_norm(void* stream, int epi, const void* X_pk, const void* W_pk, int M, int N, int K, float* h_tiled, const float* gain,
                  void* out_pk, float* part, int part_ld, const float* rs_part, int rs_nparts, float rs_inv_d, float rs_eps) {
    if ((K & 63) || (N & 31) || (M & 31)) return MG_E_SHAPE;
    GemmArgs a{};
    a.X = (const uint16_t*)X_pk; a.W = (const uint16_t*)W_pk; a.M = M; a.N = N; a.K = K;
    a.out_f32 = h_tiled; a.gain = gain; a.out_pk = (uint16_t*)out_pk; a.part = part; a.ldo = part_ld;
    a.rs = RowScale{rs_part, rs_nparts, rs_inv_d, rs_eps};
    gemm(a, epi, (mgStream_t)stream);
    return MG_OK;
}

// Large-M GEMM restricted to the 32-row tiles with a non-zero entry in row_mask [M] (M a multiple of 32): the row-tile list form
// the encoder uses.  list_scratch: M/32 + 1 ints.  epi as mgk_gemm_norm (5: tiled residual + packed x + partial sums; 2 / 3: packed
// outputs, rows scaled by rs_*) or 0 / 1 (fp32 store / accumulate, row-major ldo = N).
int mgk_gemm_row_tiles(void* stream, int epi, const void* X_pk, const void* W_pk, int M, int N, int K, float* out_f32, const float* gain,
                       void* out_pk, float* part, int part_ld, const float* rs_part, int rs_nparts, float rs_inv_d, float rs_eps,
                       const uint8_t* row_mask, int* list_scratch) {
    if ((K & 63) || (N & 31) || (M & 31) || !row_mask || !list_scratch) return MG_E_SHAPE;
    row_tile_list(row_mask, M, list_scratch + 1, list_scratch, (mgStream_t)stream);
    GemmArgs a{};
    a.X = (const uint16_t*)X_pk; a.W = (const uint16_t*)W_pk; a.M = M; a.N = N; a.K = K;
    a.out_f32 = out_f32; a.gain = gain; a.out_pk = (uint16_t*)out_pk; a.part = part; a.ldo = (epi == EPI_RESID_NORM) ? part_ld : N;
    a.rs = RowScale{rs_part, rs_nparts, rs_inv_d, rs_eps};
    a.row_tiles = list_scratch + 1; a.n_row_tiles = list_scratch;
    gemm(a, epi, (mgStream_t)stream);
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

int mgk_attention(void* stream, int mode, const void* Q, const void* K, const void* Vt, void* ctx_pk, int B, int H,
                  int Sq, int Sk, int Sq_cap, int Sk_cap, const uint8_t* kmask, const float* tab1, int tab1_len,
                  const float* tabh, const float* tabv, const double* cx, const double* cy, const int* bk1, const int* bkhv,
                  void* bidx_scratch) {
    if ((Sq_cap & 31) || (Sk_cap & 63) || Sk > Sk_cap || Sq > Sq_cap || mode < 0 || mode > 2) return MG_E_SHAPE;
    AttnArgs a{};
    a.Q = (const uint16_t*)Q; a.K = (const uint16_t*)K; a.Vt = (const uint16_t*)Vt; a.ctx = (uint16_t*)ctx_pk;
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.Sq_cap = Sq_cap; a.Sk_cap = Sk_cap; a.mode = mode; a.kmask = kmask;
    a.tab1 = tab1; a.tab1_len = tab1_len; a.tabh = tabh; a.tabv = tabv;
    if (mode == ATT_ENC) {
        if (!bidx_scratch || !bk1 || !bkhv || !cx || !cy || Sq_cap != Sk_cap) return MG_E_ARG;
        bias_index((uint16_t*)bidx_scratch, cx, cy, kmask, bk1, bkhv, B, Sk, Sk_cap, (mgStream_t)stream);
        a.bidx = (const uint16_t*)bidx_scratch;
        a.bk1 = bk1;
    }
    attention(a, (mgStream_t)stream);
    return MG_OK;
}

// encoder attention with the padded-stage / padded-block skip lists built from the key mask (as mg_encode runs it)
int mgk_attention_enc_skip(void* stream, const void* Q, const void* K, const void* Vt, void* ctx_pk, int B, int H, int S, int S_cap,
                           const uint8_t* kmask, const float* tab1, const float* tabh, const float* tabv, const double* cx,
                           const double* cy, const int* bk1, const int* bkhv, void* bidx_scratch, int* kst_scratch,
                           uint8_t* qbv_scratch) {
    if ((S_cap & 63) || S > S_cap || !kmask || !bidx_scratch || !kst_scratch || !qbv_scratch) return MG_E_ARG;
    AttnArgs a{};
    a.Q = (const uint16_t*)Q; a.K = (const uint16_t*)K; a.Vt = (const uint16_t*)Vt; a.ctx = (uint16_t*)ctx_pk;
    a.B = B; a.H = H; a.Sq = S; a.Sk = S; a.Sq_cap = S_cap; a.Sk_cap = S_cap; a.mode = ATT_ENC; a.kmask = kmask;
    a.tab1 = tab1; a.tab1_len = 32; a.tabh = tabh; a.tabv = tabv;
    bias_index((uint16_t*)bidx_scratch, cx, cy, kmask, bk1, bkhv, B, S, S_cap, (mgStream_t)stream);
    attn_lists(kmask, B, S, S_cap, kst_scratch, qbv_scratch, (mgStream_t)stream);
    a.bidx = (const uint16_t*)bidx_scratch; a.bk1 = bk1; a.kst = kst_scratch; a.qbv = qbv_scratch;
    attention(a, (mgStream_t)stream);
    return MG_OK;
}

int mgk_attention_step(void* stream, const void* q, const void* Kc, const void* Vc, void* ctx_pk, int rows, int H,
                       int group, int cap, const int* len, int n_keys, const float* bias, const int* anc, int t) {
    if (group < 1 || group > 8) return MG_E_SHAPE;
    AttnStepArgs a{};
    a.q = (const uint16_t*)q; a.Kc = (const uint16_t*)Kc; a.Vc = (const uint16_t*)Vc; a.ctx = (uint16_t*)ctx_pk;
    a.rows = rows; a.H = H; a.group = group; a.cap = cap; a.len = len; a.n_keys = n_keys; a.bias = bias; a.anc = anc;
    a.t = t;
    attention_step(a, (mgStream_t)stream);
    return MG_OK;
}

#ifdef MG_TOOLS
int mgk_attention_step_trace(void* stream, const void* q, const void* Kc, const void* Vc, void* ctx_pk, int rows, int H, int cap,
                             const int* len, long long* trace) {
    AttnStepArgs a{};
    a.q = (const uint16_t*)q; a.Kc = (const uint16_t*)Kc; a.Vc = (const uint16_t*)Vc; a.ctx = (uint16_t*)ctx_pk;
    a.rows = rows; a.H = H; a.group = 1; a.cap = cap; a.len = len;
    attention_step_trace(a, trace, (mgStream_t)stream);
    return MG_OK;
}
#endif

// Weight-absorbed cross-attention of the greedy decode step (k_xattn.hip), the three launches of a layer + the weight re-ordering:
// q [rows][H][64] bf16, wkv fp32 [2*H*64][d] (K rows first), enc [owners][cap][d] bf16 natural rows, len [owners], kv_owner [rows]
// (nullable).  Scratch: wk, wv [H*d*64] bf16 each, qx [rows][H][d] bf16, part [rows][nsplit][H][d] bf16, ml [rows][nsplit][H][2] fp32.
// Result: ctx_pk packed [rows padded to 32][H*64] bf16.
int mgk_xattn(void* stream, const void* q, const float* wkv, const void* enc, const int* len, const int* kv_owner, int rows, int H, int d,
              int cap, int nsplit, int nstg, void* wk, void* wv, void* qx, void* part, float* ml, void* ctx_pk) {
    if (!xattn_supported(d, H) || nsplit < 1 || nsplit > 4 || (nstg != 3 && nstg != 4)) return MG_E_UNSUPPORTED;
    mgStream_t st = (mgStream_t)stream;
    xattn_pack_weights(wkv, (uint16_t*)wk, (uint16_t*)wv, H, d, st);
    xattn_stream_prepare(d, nstg);
    XAttnArgs a{};
    a.q = (const uint16_t*)q; a.qx = (uint16_t*)qx; a.wk = (const uint16_t*)wk; a.wv = (const uint16_t*)wv; a.enc = (const uint16_t*)enc;
    a.len = len; a.kv_owner = kv_owner; a.part = (uint16_t*)part; a.ml = ml; a.ctx = (uint16_t*)ctx_pk;
    a.rows = rows; a.H = H; a.d = d; a.cap = cap; a.nsplit = nsplit; a.nstg = nstg;
    { const char* e = getenv("MG_XATTN_NT"); a.nt = e ? atoi(e) : 1; }
    xattn_expand(a, st);
    xattn_stream(a, st);
    xattn_contract(a, st);
    return MG_OK;
}
// rows of a packed [B*rows_per_image][d] bf16 operand -> natural rows dst[b][row_map[r]][d] (k_xattn.hip enc_rows)
int mgk_enc_rows(void* stream, const void* src_pk, const int* row_map, void* dst, int B, int rows_per_image, int cap, int d) {
    enc_rows((const uint16_t*)src_pk, row_map, (uint16_t*)dst, B, rows_per_image, cap, d, (mgStream_t)stream);
    return MG_OK;
}

size_t mgk_embed_meta_bytes(int B, int S_cap) { return embed_meta_bytes(B, S_cap); }

int mgk_embed_assemble(void* stream, void* meta_ws, const int64_t* input_ids, const float* bbox,
                       const uint8_t* attention_mask, const float* patch_emb, const void* tok_emb, const void* x_emb,
                       const void* y_emb, int B, int L, int P, int d, int n_side, int M2, int V, int S_cap,
                       float* hidden, double* cx, double* cy, uint8_t* mask, int* xrow, int* xlen, int* err) {
    if (S_cap < L + P || (d & 3) || P != n_side * n_side) return MG_E_SHAPE;
    EmbedArgs a{};
    a.input_ids = input_ids; a.bbox = bbox; a.attn_mask = attention_mask; a.patch_emb = patch_emb;
    a.tok_emb = (const uint16_t*)tok_emb; a.x_emb = (const uint16_t*)x_emb; a.y_emb = (const uint16_t*)y_emb;
    a.B = B; a.L = L; a.P = P; a.d = d; a.n_side = n_side; a.M2 = M2; a.V = V; a.S_cap = S_cap;
    a.hidden = hidden; a.cx = cx; a.cy = cy; a.mask = mask; a.xrow = xrow; a.xlen = xlen; a.err = err;
    embed_assemble(a, meta_ws, (mgStream_t)stream);
    return MG_OK;
}

int mgk_greedy_select(void* stream, const float* logits, int rows, int V, int ldl, int eos, int pad, int min_len,
                      int64_t* next_ids, int64_t* out_ids, int max_len, int pos, int* unfinished, int* n_unfinished,
                      float* top2) {
    ArgmaxArgs a{};
    a.logits = logits; a.rows = rows; a.V = V; a.ldl = ldl; a.eos = eos; a.pad = pad; a.min_len = min_len;
    a.next_ids = next_ids; a.out_ids = out_ids; a.max_len = max_len; a.pos = pos; a.unfinished = unfinished;
    a.n_unfinished = n_unfinished; a.top2 = top2;
    mg_memset_async(n_unfinished, 0, sizeof(int), (mgStream_t)stream);   // the kernel accumulates
    greedy_select(a, (mgStream_t)stream);
    return MG_OK;
}

int mgk_gemm_splitk(void* stream, const void* X_pk, const void* W_pk, float* P, int M, int N, int K, int ldp,
                    size_t slab_stride, int KS) {
    if ((K & 63) || KS < 1 || KS > 16 || M > 256) return MG_E_SHAPE;
    RowScale rs{};
    gemm_rows_splitk((const uint16_t*)X_pk, (const uint16_t*)W_pk, P, M, N, K, ldp, slab_stride, KS, rs, (mgStream_t)stream);
    return MG_OK;
}
int mgk_splitk_factor(int N, int K) { return splitk_factor(N, K); }
int mgk_gemm_set_variant(int v) { gemm_set_variant(v); return MG_OK; }

int mgk_gemm_resid(void* stream, const void* X_pk, const void* W_pk, float* h, const float* gain, float gscale, void* x_pk,
                   float* part, int M, int N, int K, const float* rs_part, int rs_nparts, float rs_inv_d, float rs_eps) {
    if ((K & 63) || (N & 31) || M > 256) return MG_E_SHAPE;
    ResidArgs r{};
    r.X = (const uint16_t*)X_pk; r.W = (const uint16_t*)W_pk; r.h = h; r.gain = gain; r.gscale = gscale; r.x_pk = (uint16_t*)x_pk;
    r.part = part; r.M = M; r.N = N; r.K = K; r.rs = RowScale{rs_part, rs_nparts, rs_inv_d, rs_eps};
    gemm_rows_resid(r, (mgStream_t)stream);
    return MG_OK;
}

int mgk_gemm_resid_mt(void* stream, const void* X_pk, const void* W_pk, float* h, const float* gain, float gscale, void* x_pk,
                      float* part, int M, int N, int K, const float* rs_part, int rs_nparts, float rs_inv_d, float rs_eps, int wide_tiles,
                      float* kpart, int* ticket) {
    if ((K & 63) || (N & 31) || M > 256) return MG_E_SHAPE;
    ResidArgs r{};
    r.X = (const uint16_t*)X_pk; r.W = (const uint16_t*)W_pk; r.h = h; r.gain = gain; r.gscale = gscale; r.x_pk = (uint16_t*)x_pk;
    r.part = part; r.M = M; r.N = N; r.K = K; r.rs = RowScale{rs_part, rs_nparts, rs_inv_d, rs_eps};
    r.wide_tiles = wide_tiles; r.kpart = kpart; r.ticket = ticket;
    gemm_rows_resid(r, (mgStream_t)stream);
    return MG_OK;
}
int mgk_set_rows_mt(int on) { gemm_rows_set_mt(on); return MG_OK; }
int mgk_set_attention_qt(int qt) { attention_set_qt(qt); return MG_OK; }
int mgk_set_pp_parts(int mode) { gemm_pp_set_parts(mode); return MG_OK; }

#ifdef MG_TOOLS
int mgk_gemm_resid_trace(void* stream, const void* X_pk, const void* W_pk, float* h, const float* gain, void* x_pk, float* part, int N,
                         int K, const float* rs_part, long long* trace) {
    ResidArgs r{};
    r.X = (const uint16_t*)X_pk; r.W = (const uint16_t*)W_pk; r.h = h; r.gain = gain; r.gscale = 1.0f; r.x_pk = (uint16_t*)x_pk;
    r.part = part; r.M = 32; r.N = N; r.K = K; r.rs = RowScale{rs_part, N / 8, 1.0f / (float)N, 1e-6f};
    gemm_rows_resid_trace(r, trace, (mgStream_t)stream);
    return MG_OK;
}
#endif

int mgk_gemm_pair(void* stream, const void* Wn_pk, const void* Wr_pk, const float* gain, int N2, int d, int inner, void* W2_pk,
                  float* scratch_f32, const void* xwin_pk, float* h, void* hb_out_pk, float* part, void* out2_pk, int M, int relu) {
    if ((d & 63) || (inner & 63) || (N2 & 31) || M > 256) return MG_E_SHAPE;
    mgStream_t st = (mgStream_t)stream;
    const int K2 = d + inner;
    float *A = scratch_f32, *Bm = A + (size_t)N2 * d, *Cm = Bm + (size_t)d * inner;
    unpack_weight((const uint16_t*)Wr_pk, Bm, d, inner, st);
    unpack_weight((const uint16_t*)Wn_pk, A, N2, d, st);
    scale_cols_f32(A, gain, Cm, N2, d, K2, st);
    gemm_f32_scaled(A, gain, Bm, Cm + d, N2, d, inner, K2, st);
    pack_weight(Cm, 0, N2, K2, (uint16_t*)W2_pk, N2, st);
    ResidArgs r{};
    r.X = (const uint16_t*)xwin_pk; r.x_kts = K2 >> 4; r.x_k0 = d >> 4; r.W = (const uint16_t*)Wr_pk; r.h = h;
    r.x2_pk = (uint16_t*)hb_out_pk; r.part = part; r.M = M; r.N = d; r.K = inner;
    GemmArgs g{};
    g.X = (const uint16_t*)xwin_pk; g.W = (const uint16_t*)W2_pk; g.M = M; g.N = N2; g.K = K2; g.out_pk = (uint16_t*)out2_pk;
    if (!relu) return MG_E_UNSUPPORTED;      // the per-head form is covered through mg_generate
    gemm_rows_pair(r, g, EPI_PK_RELU, st);
    return MG_OK;
}

int mgk_add_norm_pack(void* stream, float* h, const float* P, int KS, int ldp, size_t slab_stride, const float* gain,
                      void* x_pk, int M, int d, float eps, float scale) {
    if ((d & 15) || d > 4096 || KS < 0 || KS > 16) return MG_E_SHAPE;
    Slabs sl; sl.P = P; sl.KS = KS; sl.ldp = ldp; sl.stride = slab_stride;
    add_norm_pack(h, sl, gain, (uint16_t*)x_pk, M, d, eps, scale, (mgStream_t)stream);
    return MG_OK;
}

int mgk_relu_pack(void* stream, const float* P, int KS, int ldp, size_t slab_stride, void* y_pk, int M, int N) {
    if ((N & 15) || KS < 1 || KS > 16) return MG_E_SHAPE;
    Slabs sl; sl.P = P; sl.KS = KS; sl.ldp = ldp; sl.stride = slab_stride;
    relu_pack(sl, (uint16_t*)y_pk, M, N, (mgStream_t)stream);
    return MG_OK;
}

size_t mg_preprocess_scratch_bytes(int B, int Hs, int Ws, int out_size) {
    if (B < 1 || Hs < 1 || Ws < 1 || out_size < 1) return 0;
    return preprocess_scratch_bytes(B, Hs, Ws, out_size);
}

int mg_preprocess_pages(void* stream, const uint8_t* pages_u8, int B, int Hs, int Ws, int out_size, float* pixel_values,
                        void* scratch, size_t scratch_bytes) {
    if (!pages_u8 || !pixel_values || !scratch || B < 1 || Hs < 1 || Ws < 1 || out_size < 1) return MG_E_ARG;
    if (scratch_bytes < preprocess_scratch_bytes(B, Hs, Ws, out_size)) return MG_E_WORKSPACE;
    preprocess_pages(pages_u8, B, Hs, Ws, out_size, pixel_values, scratch, (mgStream_t)stream);
    return MG_OK;
}

}  // extern "C"
