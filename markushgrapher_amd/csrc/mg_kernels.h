// Host-side launchers for the HIP kernels of the MarkushGrapher-2 VTL encoder / CXSMILES decoder path.
// Every launcher only enqueues work on `stream`; no allocation, no synchronisation.
#pragma once
#include "mg_device.h"

namespace mg {

// ---------------------------------------------------------------------------------------------
// GEMM  out[m][n] = sum_k X[m][k] * W[n][k]   (X, W bf16 in the packed fragment-tile format, fp32 accumulate)
// ---------------------------------------------------------------------------------------------
enum GemmEpi : int {
    EPI_F32_STORE = 0,   // out_f32[m*ldo + n] = acc (+ bias[n])                       (patch embed, logits)
    EPI_F32_RESID = 1,   // out_f32[m*ldo + n] += acc                                   (attention O, FFN wo)
    EPI_PK_RELU = 2,     // out_pk packed [M][N] = bf16(relu(acc))                      (FFN wi)
    EPI_PK = 3,          // out_pk packed [M][N] = bf16(acc)
    EPI_HEADS = 4,       // per-head Q/K/V targets, see HeadsOut                        (QKV and cross-K/V projections)
    // Encoder residual projection with the NEXT RMSNorm folded in (deferred norm, as in the decode step):
    //   h (fp32, TILED layout ht_off) += acc;  out_pk = pack(bf16(h * gain)) un-normalised;  part[m][N/64] = sums of h^2 over
    //   each wave's 64 columns.  Consumers (EPI_HEADS / EPI_PK_RELU with GemmArgs::rs set) multiply their output rows by
    //   rsqrt(sum(part[m]) / N + eps): RMSNorm is a per-row scalar, so no norm launch sits between the GEMMs.
    EPI_RESID_NORM = 5,
    // decode-step kernels only (gemm_rows): W rows interleaved gate_0, up_0, gate_1, up_1, ...; out_pk packed [M][N/2] =
    // bf16(silu(r g_j) * (r u_j)) with r the deferred RMSNorm scale of the row (Llama MLP, modeling_llama.py LlamaMLP)
    EPI_PK_SWIGLU = 6,
    // large-M tile kernel only (gemm_has_gelu_epilogue): out_pk packed [M][N] = bf16(gelu_tanh(acc + bias[n])), bias null = 0   (vision-tower MLP, SiglipMLP)
    EPI_PK_GELU = 7,
    // OCSR vision branch (Swin, swin.hip): projections with a bias vector.  out_pk packed [M][N] = bf16(acc + bias[n]) and
    // bf16(gelu_erf(acc + bias[n])) (exact erf form: SwinMLP / hidden_act "gelu", stock modeling_swin.py:471-483); bias null = 0
    EPI_PK_BIAS = 8,
    EPI_PK_GELU_ERF = 9,
};
// Destination formats for per-head projections (head dim fixed at 64):
enum HeadFmt : int {
    HF_NONE = 0,
    HF_PK_ROWS = 1,   // [B][H][S_cap/32][4][512]  rows = token, k = head dim   (MFMA operand for Q·K^T)
    HF_PK_T = 2,      // [B][H][2][S_cap/16][512]  rows = head dim, k = token   (MFMA operand for P·V)
    HF_NATURAL = 3,   // [B][H][S_cap][64]         row = row_map[token] (or token)  (single-query decode streams)
    HF_STEP_Q = 4,    // decode step: token m is sequence row m -> [rows][H][64]
    HF_STEP_KV = 5,   // decode step: -> cache[row][H][S_cap][64] at position `pos`
};
struct HeadsOut {
    uint16_t* ptr[3];      // target of column region n / inner
    int fmt[3];
    int inner;             // H * 64
    int H;
    int S_in;              // tokens per batch item in the GEMM's row space (m = b*S_in + s)
    int S_cap;             // token capacity of the destination per (b,h)
    const int* row_map;    // HF_NATURAL: destination row of token m, or -1 to drop (nullable)
    int s_off;             // HF_PK_ROWS / HF_PK_T: destination token index = s + s_off (key blocks laid side by side)
    int pos;               // HF_STEP: cache position written
    const int* pos_dev;    // if non-null the position is read from device memory (graph replay)
    const int* pos_rows;   // if non-null: per-row positions [rows] (continuous decoding: every slot is at its own position)
};
// Deferred RMSNorm statistic of the decode step: the activations X were stored UN-normalised (bf16(h * gain)) by the
// previous residual projection, which also left per-row partial sums of squares; since RMSNorm is a per-row scalar,
// the consumer GEMM multiplies its outputs by r(m) = rsqrt(sum_i part[m*nparts + i] * inv_d + eps) instead.
struct RowScale {
    const float* part;     // null = X is already normalised (scale 1)
    int nparts;            // multiple of 4
    float inv_d, eps;
};
struct GemmArgs {
    const uint16_t* X;
    const uint16_t* W;
    int M, N, K;
    float* out_f32;
    int ldo;
    const float* bias;
    uint16_t* out_pk;
    HeadsOut heads;
    RowScale rs;           // deferred RMSNorm scale of the rows of X (decode-step kernels; encoder: EPI_HEADS / EPI_PK_RELU)
    const float* gain;     // EPI_RESID_NORM: gain of the next RMSNorm (null: no packed output / partial sums)
    float* part;           // EPI_RESID_NORM: [M][N/64] partial sums of squares
    // decode-step kernels: X may be a column window of a wider packed buffer: x_kts = 16-wide k-tiles per row tile of
    // the buffer (0 = K/16), x_k0 = first k-tile of the window
    int x_kts, x_k0;
    // EPI_PK_SWIGLU: the packed output as a column window of a wider buffer (out_ld columns, first column out_col0); out_ld = 0: N/2, 0
    int out_ld, out_col0;
    // Large-M tile kernel (320x256 / 256x256) only, optional: compute only the 32-row tiles listed in row_tiles[0 .. *n_row_tiles)
    // (ascending tile ids; rows of other tiles are neither read nor written).  The encoder passes the tiles that hold at least one
    // attended position: padded text slots and the slots of dropped patches form whole dead tiles (16 % of the rows at the
    // benchmark's length distribution).  Kernels without list support ignore it and compute every row - the list is an optimisation.
    const int* row_tiles;
    const int* n_row_tiles;
    int both_halves;       // decode-step half-tile projections with >= 3 row tiles: 1 = a workgroup takes both 16-feature halves of its weight tile (same bits;
                           // half the activation re-reads through L2: better beside other contexts, slower for a call alone)
    int pp_colgroup;       // set by the ping-pong kernel's launcher: > 0: an XCD's tiles ordered column group by column group (that many column tiles wide)
    int pp_balance;        // set by the ping-pong kernel's launcher: cut the row tiles into blocks of balanced height (k_gemm_pp.hip)
    int pp_part, pp_parts; // set by the ping-pong kernel's launcher: this launch computes part pp_part of pp_parts of the row tiles (list entries or tiles
                           // [part * n / parts, (part + 1) * n / parts)): a problem with more tiles per workgroup than the kernel's table holds runs as several launches
};
void gemm(const GemmArgs& a, int epi, mgStream_t stream);
bool gemm_has_gelu_epilogue(int M, int N);     // EPI_PK_GELU exists in the 320x256 / 256x256 tile kernels only
bool gemm_pp(const GemmArgs& a, int epi, int ti, mgStream_t stream);   // k_gemm_pp.hip: persistent ping-pong tile kernel (ti = 4, 5); false = shape not supported
void gemm_pp_set_parts(int mode); // ping-pong kernel: row tiles of one problem over several launches - 0 (default) never, 1 when its tile table is too small, 2 .. 8 always that many (tests; same bits)
void gemm_set_variant(int v);   // 0: 128x128 kernel only; 1: + 256x128 three-stage; 2: + 256x256; 4: + 320x256 wherever it fits; 3 (default): by shape

// small-M (decode step) GEMM: M <= 32*MT rows of live sequences, weights streamed once.
void gemm_rows(const GemmArgs& a, int epi, mgStream_t stream);
void gemm_rows_set_resid_f16(int on);  // 1 (default): residual projections with several row tiles take 16 features per workgroup; 0: 8 (tests, A/B runs; same bits)
void gemm_rows_set_mt(int on);   // 1: K-slab form (default 0: measured slower, k_gemm.hip) of the projections with several row tiles where the caller provides kpart / ticket; 0: one-workgroup forms (same bits)
void gemm_rows_set_ft2(int mode);  // -1 (default): GemmArgs::both_halves decides; 0 / 1: never / always both 16-feature halves of a weight tile per workgroup (tests, A/B runs; same bits)
void gemm_rows_set_split(int mode);   // row-tile split policy of the decode projections (-1 default by weight size, 0 never, 1 always one tile per workgroup)

// split-K decode GEMM: P[ks][m*ldp + n] (ks < KS) = partial sums over the ks-th K range; consumers add the slabs
int splitk_factor(int N, int K);
// lm_head form with the greedy selection started in the epilogue (KS = 1): every workgroup (32 output features) also leaves, per
// row, the top-2 of ITS features - ptop[(m * ntiles + nt)] = {best value, second value, index of the best (int bits), -} - with the
// stop tokens stop[0..3] (-1 = unused) left out of the ranking and their logits stored apart in stopv[m][4], so that the selection
// kernel can apply MinLength suppression without the step position being known here.  The selection then reduces N/32 partials per
// row instead of N logits.  write_logits = 0: the fp32 logits are not stored at all.
struct TopOut {
    float4* ptop;          // [M][ceil(N/32)]; null = plain projection
    float* stopv;          // [M][4]
    int stop[4];
    int write_logits;
};
void gemm_rows_splitk(const uint16_t* X, const uint16_t* W, float* P, int M, int N, int K, int ldp, size_t slab_stride, int KS,
                      const RowScale& rs, mgStream_t stream, const TopOut* top = nullptr);
// Residual projection of the decode step with the NEXT sub-layer's RMSNorm folded in (no separate norm launch):
//   h[m][n] += sum_k X[m][k] W[n][k];   x_pk = pack(bf16(h * gain * gscale))  (un-normalised);
//   part[m*(N/8) + n/8] = sum over the block's 8 features of h^2  (consumers turn them into r(m), see RowScale).
// One workgroup per 8 output features (complete sums, deterministic), K split over the workgroup's waves.
void gemm_rows_resid(const uint16_t* X, const uint16_t* W, float* h, const float* gain, float gscale, uint16_t* x_pk, float* part,
                     int M, int N, int K, const RowScale& rs, mgStream_t stream);
// general form: X as a column window (x_kts / x_k0 as in GemmArgs); two optional packed bf16 outputs, each a column
// window (ld = columns of the destination buffer, col0 = first column): x_pk = bf16(h * gain * gscale), x2_pk = bf16(h)
struct ResidArgs {
    const uint16_t* X;
    int x_kts, x_k0;
    const uint16_t* W;
    float* h;
    const float* gain;
    float gscale;
    uint16_t* x_pk;
    int x_ld, x_col0;      // x_ld = 0: N
    uint16_t* x2_pk;
    int x2_ld, x2_col0;
    float* part;
    int M, N, K;
    RowScale rs;
    int alone;             // 1: the call has the GPU to itself (no other execution context beside it): forms that are faster alone but move more bytes
                           // may be taken - the K-slab form in two launches from 4 row tiles on (k_gemm.hip); same bits either way
    int wide_tiles;        // row tiles up to which the long K = d_ff form keeps 16 K-partitioning waves (0: 4).  The decode steps pass 8 (every call
                           // the C ABI admits: 256 rows), so that a row sums in the order of a 32-row call whatever its call's size
    // Several row tiles, K-slab form (gemm_rows_resid_mt_kernel): the K chunks that the waves of the one-workgroup forms take become
    // workgroups - kpart [chunks][rows padded to 32][N] fp32 partial sums, ticket [N / 32] zero-initialised arrival counters (left at
    // zero by every launch).  Both null: the one-workgroup forms.
    float* kpart;
    int* ticket;
};
void gemm_rows_resid(const ResidArgs& r, mgStream_t stream);
void gemm_rows_resid_trace(const ResidArgs& r, long long* trace, mgStream_t stream);   // phase stamps, M <= 32, K = d_ff form
// Two independent decode projections that read the same inputs in ONE launch (they sit side by side in the grid):
// the residual projection `r` and the projection `g` (epilogue EPI_HEADS or EPI_PK_RELU, half-tile workgroups).
// Used with product weights: g's X is the window [bf16(h_before) | ctx] and its W = [Wn·G | Wn·G·Wr], so that
// g = Wn·G·(h_before + Wr·ctx) = Wn·G·h_after without waiting for r's result (see engine.hip, decode step).
void gemm_rows_pair(const ResidArgs& r, const GemmArgs& g, int epi, mgStream_t stream);
// fp32 helpers for building product weights at finalize time
void unpack_weight(const uint16_t* W_pk, float* out, int N, int K, mgStream_t stream);          // out[n][k] row-major
// C[n][j] (ldc) = sum_k A[n][k] * gain[k] * B[k][j]   (A: [N][K] row-major, B: [K][J] row-major, fp32)
void gemm_f32_scaled(const float* A, const float* gain, const float* B, float* C, int N, int K, int J, int ldc, mgStream_t stream);
// C[n][j] (ldc) = A[n][j] * gain[j]
void scale_cols_f32(const float* A, const float* gain, float* C, int N, int J, int ldc, mgStream_t stream);
// a row-major fp32 matrix given as KS split-K partial slabs
struct Slabs {
    const float* P;        // null = not used
    int KS, ldp;
    size_t stride;         // elements between slabs
};

// ---------------------------------------------------------------------------------------------
// normalisation / packing
// ---------------------------------------------------------------------------------------------
// x_pk[M][d] (packed bf16) = RMSNorm(h[M][d] fp32) * gain * scale ; optionally also out_f32 row-major
void rmsnorm_pack(const float* h, const float* gain, uint16_t* x_pk, float* out_f32, int M, int d, float eps,
                  float scale, mgStream_t stream);
// the same from the TILED fp32 layout (ht_off) of the encoder's residual stream; M a multiple of 32
void rmsnorm_pack_tiled(const float* h_tiled, const float* gain, uint16_t* x_pk, float* out_f32, int M, int d, float eps,
                        mgStream_t stream);
// same, but source row m is written to packed row dst_row[m] (skipped when negative)
void rmsnorm_pack_rows(const float* h, const float* gain, uint16_t* x_pk, const int* dst_row, int M, int d, float eps,
                       float scale, mgStream_t stream);
// h[m][:] += sum_s add.P[s][m][:] (fixed order), then x_pk = pack(RMSNorm(h) * gain * scale)   (decode-step fusion of
// the residual add of the previous projection with the next sub-layer's norm)
void add_norm_pack(float* h, const Slabs& add, const float* gain, uint16_t* x_pk, int M, int d, float eps, float scale,
                   mgStream_t stream);
// y_pk[M][N] packed = bf16(relu(sum_s in.P[s]))
void relu_pack(const Slabs& in, uint16_t* y_pk, int M, int N, mgStream_t stream);
// HF [N][K] weight (fp32 or bf16 bits) -> packed bf16 tiles, rows >= N zero
void pack_weight(const void* src, int src_is_bf16, int N, int K, uint16_t* dst, int Npad, mgStream_t stream);
void convert_to_f32(const void* src, int src_is_bf16, float* dst, size_t n, mgStream_t stream);
void convert_to_bf16(const void* src, int src_is_bf16, uint16_t* dst, size_t n, mgStream_t stream);
// OCSR-branch embeddings e1 [B][M][d] fp32 (precomputed by the caller) -> packed bf16 rows [B * M_pad][d]
// (rows j >= M zero), row_map [B * M_pad] (j < M ? j : -1) for the compacted cross K/V stream, and - when xmask is
// given - the key mask of the teacher-forced cross-attention [B][M_pad + S_cap] = [1 x M | 0 | enc_mask]
void pack_e1(const float* e1, int B, int M, int M_pad, int d, uint16_t* e1_pk, int* row_map, const uint8_t* enc_mask, int S_cap,
             uint8_t* xmask, mgStream_t stream);
// pixel_values [B][C][I][I] fp32 -> packed bf16 im2col matrix [B*P][C*ps*ps]
void im2col_pack(const float* pix, uint16_t* x_pk, int B, int C, int I, int ps, mgStream_t stream);

// ---------------------------------------------------------------------------------------------
// encoder input assembly (combine_image_text_embeddings + cell embedding)
// ---------------------------------------------------------------------------------------------
struct EmbedArgs {
    const int64_t* input_ids;   // [B][L]
    const float* bbox;          // [B][L][4]
    const uint8_t* attn_mask;   // [B][L] or null (null = everything attended, incl. visual padding)
    const float* patch_emb;     // [B][P][d] fp32
    const uint16_t* tok_emb;    // [V][d] bf16
    const uint16_t* x_emb;      // [M2][d] bf16
    const uint16_t* y_emb;      // [M2][d] bf16
    int B, L, P, d, n_side, M2, V, S_cap;
    float* hidden;              // [B][S_cap][d] row-major, or the tiled layout ht_off when hidden_tiled != 0
    int hidden_tiled;
    double* cx;                 // [B][S_cap]  box x-centre (float64 as in the reference)
    double* cy;
    uint8_t* mask;              // [B][S_cap]  1 = attended
    int* xrow;                  // [B][S_cap]  compacted cross-attention row of token s, -1 if masked
    int* xlen;                  // [B]         x_row0 + number of attended tokens
    int x_row0;                 // first compacted row (rows [0, x_row0) of the cross K/V stream belong to the e1 tokens)
    // Trailing text padding (attn_mask given): 0 = stock batched semantics - padded slots stay between the text and the patches, so
    // UDOP's 1-D position bias counts them; 1 = per-image semantics - the patches follow the image's LAST ATTENDED text token, the
    // padded slots move behind the visual block: every image is computed as if it were alone and unpadded (the reference's batch
    // size is 1).  text_len [B] (nullable) receives the per-image text length either way (L in mode 0).
    int trim_padding;
    int* text_len;
    int* err;                   // device error word (bit 0: token id out of range)
};
size_t embed_meta_bytes(int B, int S_cap);
void embed_assemble(const EmbedArgs& a, void* meta_ws, mgStream_t stream);

// ---------------------------------------------------------------------------------------------
// attention over packed Q/K/V^T (encoder self-attention, teacher-forced decoder self- and cross-attention)
// ---------------------------------------------------------------------------------------------
enum AttnMode : int { ATT_ENC = 0, ATT_DEC_SELF = 1, ATT_CROSS = 2 };
struct AttnArgs {
    const uint16_t* Q;      // HF_PK_ROWS [B][H][Sq_cap/32][4][512]
    const uint16_t* K;      // HF_PK_ROWS [B][H][Sk_cap/32][4][512]
    const uint16_t* Vt;     // HF_PK_T    [B][H][2][Sk_cap/16][512]
    uint16_t* ctx;          // packed [B*Sq_cap][H*64]
    int B, H, Sq, Sk, Sq_cap, Sk_cap;
    int mode;
    const uint8_t* kmask;   // [B][Sk_cap] 1 = attended (nullable = all Sk attended)
    // ATT_ENC: tab1/tabh/tabv = RAW bucket tables [32][H] (1-D, horizontal, vertical), bk1 = distance -> 1-D bucket,
    //          bidx = per-(image, query, key) table addresses from bias_index();  ATT_DEC_SELF: tab1 = [tab1_len][H] indexed by distance i-j >= 0
    const float* tab1;
    const float* tabh;
    const float* tabv;
    const uint16_t* bidx;   // [B][Sk_cap/32][Sk_cap][32]
    const int* bk1;         // ATT_ENC: 1-D bucket of key - query for the 257 distances -128..128 (stock:422-468)
    int tab1_len;
    // ATT_ENC, optional (attn_lists): per image the 64-key stages that hold at least one attended key, and which
    // 128-query blocks hold at least one attended position; fully padded stages / blocks are skipped
    const int* kst;         // [B][1 + Sk_cap/64]: count, stage ids
    const uint8_t* qbv;     // [B][Sq_cap/128 rounded up]
    int dbg;                // diagnostics (MG_ATT_DBG): bit 0 = every stage waits for ALL outstanding copies
};
// bucket indices of the encoder's relative biases, once per batch (see k_attn.hip)
void bias_index(uint16_t* out, const double* cx, const double* cy, const uint8_t* kmask, const int* bk1, const int* bkhv, int B,
                int Sk, int S_cap, mgStream_t stream);
// stage / query-block lists for AttnArgs::kst / qbv from the key mask (one launch per batch)
void attn_lists(const uint8_t* kmask, int B, int Sk, int S_cap, int* kst, uint8_t* qbv, mgStream_t stream);
// live 32-row tiles of a [rows] key mask (rows a multiple of 32, mask 16-byte aligned) -> ascending tile ids + count (GemmArgs::row_tiles)
void row_tile_list(const uint8_t* kmask, int rows, int* list, int* count, mgStream_t stream);
void attention(const AttnArgs& a, mgStream_t stream);
void attention_set_qt(int qt);   // encoder attention: 1 (default) one 32-query tile per wave, 8 waves; 2: two tiles per wave, 4 waves (same bits, measured slower; tests, A/B runs)

// ---------------------------------------------------------------------------------------------
// decode step (single query per live sequence)
// ---------------------------------------------------------------------------------------------
struct AttnStepArgs {
    const uint16_t* q;        // [rows][H][64] bf16
    const uint16_t* Kc;       // self: [rows_phys][H][T_cap][64]; cross: [B][H][S_cap][64]
    const uint16_t* Vc;
    uint16_t* ctx;            // packed [rows_pad][H*64]
    int rows, H, group;       // group = beams sharing one cross K/V (1 for self-attention)
    int cap;                  // T_cap or S_cap
    const int* len;           // per K/V owner: number of keys (cross: xlen[b]); null = use `n_keys`
    int n_keys;               // self: t+1
    const float* bias;        // self: [T_cap][H] indexed by distance t - j (nullable)
    const int* anc;           // self with beams: [T_cap][rows] physical row holding position j (nullable)
    int t;                    // current position (self)
    const int* t_dev;         // if non-null: t (and n_keys = t+1 for self-attention) are read from device memory
    int t_off;                // added to *t_dev (the OCR stage's position = prompt length - 1 + step counter)
    RowScale qrs;             // deferred RMSNorm scale of the query rows (scores are multiplied by r(row)); part = null: none
    int ctx_ld, ctx_col0;     // ctx as a column window of a wider packed buffer (ctx_ld = 0: H*64 columns, offset 0)
    // split-K form of the projections feeding this step: q (and for self-attention k, v of the new position) are
    // given as fp32 partial slabs [rows][ldp] with q | k | v at column offsets 0 | inner | 2*inner; the kernel sums
    // them, rounds to bf16, appends k, v to the cache at position t and attends over [0, t] (self) or the cross keys.
    Slabs qkv;
    int self_append;          // 1: slabs carry q,k,v and the new position is appended; 0: slabs carry q only
    uint16_t* Kc_w;           // writable cache pointers for the append
    uint16_t* Vc_w;
    // Rotary form of the self-attention step (ChemicalOCR text model, modeling_llama.py): here H = KEY/VALUE heads and group = query
    // heads per key/value head (grouped-query attention), rows = sequences.  q, k, v of the new position come as one fp32 row
    // [rows][ld] = [H*group q heads | H k heads | H v heads] x 64, un-normalised; the kernel applies the deferred RMSNorm scale `rs`,
    // the rotation of position t (cs: [positions][64] = cos[32] | sin[32]), q * qscale, rounds to bf16, appends k, v to the cache
    // [rows][H][cap][64] and attends over [0, t] for the group's query heads in one pass over the cache.  qkv == null: not used.
    struct Rope { const float* qkv; int ld; const float* cs; RowScale rs; float qscale; } rope;
    const int* pos_rows;      // continuous decoding: the row's own position t = pos_rows[row] + t_off (self: keys [0, t], bias by t - j); overrides t / t_dev
    const int* t_off_rows;    // rotary form, nullable: per K/V pool entry (page) an offset added to the row's position (prompts of different lengths:
                              // the page's prompt length minus t_off)
    const int* kv_owner;      // continuous decoding: entry of the K/V pool that row `owner` reads - cross form: the image's stream (len is
                              // indexed by it too); rotary form: the page's own cache, which the row also appends to
    int one_wg_per_cu;        // cross form (len != null): request enough LDS that ONE workgroup of the K/V stream is resident per CU (see attention_step)
    const int* live;          // group == 1 only, nullable: rows with live[row] == 0 (finished: they emit pad whatever their
                              // logits are, gen:2927-2937) are skipped - their K/V streams are not read.
                              // INVARIANT this relies on: a skipped row's context columns keep stale values, so everything
                              // computed for that row afterwards (h, partial sums, logits, top-2 record) is meaningless and may
                              // be Inf/NaN - harmless only because every downstream kernel is ROW-INDEPENDENT (MFMA rows,
                              // per-row scales, per-row selection).  A kernel that reduces ACROSS rows must mask dead rows.
};
void attention_step(const AttnStepArgs& a, mgStream_t stream);
void attention_step_allow_shared();      // before the first launch with AttnStepArgs::one_wg_per_cu (sets the kernel's LDS limit; not inside a stream capture)
void attention_step_trace(const AttnStepArgs& a, long long* trace, mgStream_t stream);   // phase stamps, cross form, group 1

// ---- weight-absorbed cross-attention of the greedy decode step (k_xattn.hip; stock modeling_udop.py:524-575) ----
// The layer's K / V streams are replaced by ONE stream of the encoder states themselves (layer-invariant, 2·d bytes per position):
// q'_h = q_h·Wk_h (xattn_expand), scores and context against the states (xattn_stream), ctx_h = c_h·Wv_h^T (xattn_contract).
struct XAttnArgs {
    const uint16_t* q;        // [rows][H][64] bf16, un-normalised cross-attention queries (HF_STEP_Q)
    uint16_t* qx;             // [rows][H][d] bf16: q' (written by xattn_expand, read by xattn_stream)
    const uint16_t* wk;       // [H][d][64] bf16: Wk_h feature-major (xattn_pack_weights)
    const uint16_t* wv;       // [H][d/32][4][64][8] bf16: Wv_h in fragment order (xattn_pack_weights)
    const uint16_t* enc;      // [owners][cap][d] bf16: the states an image's rows attend, compacted to attended positions (enc_rows)
    const int* len;           // keys per owner
    const int* kv_owner;      // continuous decoding: pool entry read by a row (null: the row itself)
    const int* live;          // nullable: rows with live[row] == 0 are skipped (see AttnStepArgs::live)
    RowScale qrs;             // deferred RMSNorm scale of the query rows (applied to the scores)
    uint16_t* part;           // [rows][nsplit][H][d] bf16: context of a key split, normalised by the split's own sum
    float* ml;                // [rows][nsplit][H][2]: running max and sum of the split
    uint16_t* ctx;            // packed window, as AttnStepArgs::ctx
    int ctx_ld, ctx_col0;
    int rows, H, d, cap;
    int nsplit;               // key splits per row (1 .. 4): workgroups of the stream = rows * nsplit
    int nstg;                 // stages of 16 keys in the stream's LDS ring: 4 = two wave groups (136 KB), 3 = one wave group (100 KB)
    int nt;                   // 1: the stream's copies carry the non-temporal hint
};
bool xattn_supported(int d, int H);
int xattn_nf(int d);
size_t xattn_stream_lds(int d, int nstg);
void xattn_stream_prepare(int d, int nstg);    // once per width, outside any stream capture (LDS limit of the stream kernel)
void xattn_expand(const XAttnArgs& a, mgStream_t stream);
void xattn_stream(const XAttnArgs& a, mgStream_t stream);
void xattn_contract(const XAttnArgs& a, mgStream_t stream);
void xattn_pack_weights(const float* wkv_f32, uint16_t* wk, uint16_t* wv, int H, int d, mgStream_t stream);
// rows of a packed [rows][d] bf16 operand -> natural rows dst[b][row_map[r]][d] (row_map < 0: dropped)
void enc_rows(const uint16_t* src_pk, const int* row_map, uint16_t* dst, int B, int rows_per_image, int cap, int d, mgStream_t stream);
// rows [len[b], next multiple of 16) of dst[b] zeroed (the stream reads whole stages of 16 keys)
void enc_pad_rows(uint16_t* dst, const int* len, int B, int cap, int d, mgStream_t stream);

// h[rows][d] = tok_emb[ids[row]]
// embed_rows + rmsnorm_pack(h, gain, x_pk) in one launch (decode step); x2_pk (nullable) = the embedding rows, packed window
void embed_norm_rows(const int64_t* ids, const uint16_t* tok_emb, float* h, const float* gain, uint16_t* x_pk, uint16_t* x2_pk, int x2_ld,
                     int x2_col0, int rows, int d, int V, int* err, float eps, mgStream_t stream);
// optionally also x_pk (packed bf16 window, x_ld columns, first column x_col0) = the embedding rows themselves
void embed_rows(const int64_t* ids, const uint16_t* tok_emb, float* h, int rows, int d, int V, int* err,
                mgStream_t stream, uint16_t* x_pk = nullptr, int x_ld = 0, int x_col0 = 0);

// Slot table of the continuous decoder (mg_generate_stream, engine.hip): `slots` decode rows work through a queue of images; a
// row that ends (EOS or max_length) frees its slot, the refill kernel hands it the next image whose cross K/V is in the pool.
struct SlotTable {
    int* pos;          // [slots] position of the token fed to this step (0 = the start token); null = batch mode (no slot table)
    int* img;          // [slots] image decoded in the slot (row of out_ids), -1 = idle
    int* pool;         // [slots] cross K/V pool entry of that image
    int* ctr;          // stream counters (layout: engine.hip)
    int* out_len;      // [N] valid columns of every finished image
    int pool_cap;      // entries of the K/V pool (image i lives in entry i % pool_cap)
    int start_id;
    // ChemicalOCR queue form: a sequence enters a slot with the token its PREFILL selected (column 0 already written) instead of a
    // start token; one that began with a stop token (or max_len 1) is finished before it ever takes a slot
    const int64_t* first_tok;   // [N], null = start_id
    int n_stop, stop[4], max_len;
};
struct ArgmaxArgs {
    const float* logits;     // [rows][ldl]
    int rows, V, ldl;
    int eos, pad, suppress_eos;
    int n_eos_more, eos_more[3];   // further stop tokens (a list of EOS ids in generation_config.json): treated exactly as `eos`
    int64_t* next_ids;       // [rows] token fed to the next step
    int64_t* out_ids;        // [rows][max_len]
    int max_len, pos;        // column written
    const int* pos_dev;      // if non-null the column is *pos_dev + pos (graph replay: the step counter lives on the device)
                             // and `top2` is the base of a [max_len][rows][2] array indexed by that column
    int min_len;             // EOS is suppressed while pos < min_len (MinLengthLogitsProcessor)
    int* unfinished;         // [rows]
    int* n_unfinished;       // [1] accumulated here; the step-end kernel publishes and clears it
    float* top2;             // [rows][2] (nullable) top-1 / top-2 logit of this step
    // step bookkeeping folded into the selection (engine decode loop): when non-null, the LAST workgroup to finish
    // publishes the unfinished count, records the first all-finished step and advances the step counter
    // (counters layout: engine.hip); step_ctr[6] is the arrival counter
    int* step_ctr;
    SlotTable slots;         // continuous decoding: per-row positions / images (slots.pos == null: batch mode)
    // fused tail (greedy_select_fused): the logits come as per-workgroup top-2 partials of the lm_head launch (TopOut), and the
    // selected token's embedding + first RMSNorm of the next step are produced here (what embed_norm_rows does at a step's start)
    const float4* ptop;      // [rows][ntiles]
    const float* stopv;      // [rows][4] logits of the stop tokens (left out of the partials)
    int ntiles;
    const uint16_t* tok_emb; // [V][d] bf16
    float* h;                // [rows][d] fp32 residual stream of the next step
    const float* gain;       // first RMSNorm gain
    uint16_t* x_pk;          // packed bf16(RMSNorm(h) * gain)
    uint16_t* x2_pk;         // packed window: bf16(h)
    int x2_ld, x2_col0, d;
    float eps;
};
void greedy_select(const ArgmaxArgs& a, mgStream_t stream);
void greedy_select_fused(const ArgmaxArgs& a, mgStream_t stream);
// continuous decoding, after the selection of a step: idle slots take the next ready images of the queue (in slot order:
// deterministic), live count / oldest live image / step counter are published for the host
void slot_refill(const SlotTable& s, int64_t* next_ids, int* unfinished, int rows, mgStream_t stream);

// beam search on the device (k_beam.hip; restates stock generation/utils.py:3208-3525)
// queue form (continuous beam decoder): per ROW position of the token fed to the step and live flag - the K rows of an image slot
// hold the same values; pos == null: batch form (one position for all images, the batch's continue flag in counters[0])
struct BeamSlots { const int* pos; const int* live; };
size_t beam_state_bytes(int B, int K, int max_len);
void beam_init(void* state, int B, int K, int max_len, int pad, int eos, int start, int64_t* next_ids, int* anc, int T_cap,
               int* counters, mgStream_t stream);
float beam_length_divisor(int cur_len, float length_penalty);
void beam_step(void* state, const float* logits, int ldl, int V, int B, int K, int max_len, int cur_len, const int* tdev,
               const float* div_table, int eos, int min_len, float length_penalty, int early_stopping, int64_t* next_ids,
               int* beam_idx, int* counters, mgStream_t stream, const BeamSlots* slots = nullptr);
void beam_finalize(void* state, int B, int K, int max_len, int64_t* out_ids, int* out_cols, float* out_scores, mgStream_t stream);
// ancestor-table form of the KV-cache reorder (cache_utils.py:100-104): anc[j][row] <- anc[j][beam_idx[row]], j < t_written
void beam_reorder_anc(int* anc, const int* beam_idx, int rows, int t_written, const int* tdev, const int* counters, mgStream_t stream,
                      const BeamSlots* slots = nullptr);
// queue form, after the step of every slot: [end_first: stopped images are written out (out_ids [N][max_len], out_len, out_scores) and
// their slots freed, the others advance] -> idle slots are handed the next ready images of the queue (assign[slot] = image or -1) ->
// newly assigned slots get the batch form's initial state.  Counters as the greedy queue's (ctr[0] live, [1] done, [2] steps,
// [4] queue head, [5] ready, [7] oldest live image)
void beam_slots_step(void* state, int slots, int K, int max_len, int pad, int eos, int start, int early_stopping, int* pos, int* img, int* pool,
                     int* bpool, int* live, int* assign, int64_t* next_ids, int* anc, int T_cap, int pool_cap, int64_t* out_ids, int* out_len,
                     float* out_scores, int* ctr, bool end_first, mgStream_t stream);
// physical form: dst[lk][row] = src[lk][beam_idx[row]] for nlk = layers*2 K/V planes of [rows][H][t_cap][64] bf16
void beam_reorder_copy(const uint16_t* src, uint16_t* dst, const int* beam_idx, int nlk, int rows, int H, int t_cap, int t_used,
                       mgStream_t stream);

// page preprocessing (k_prep.hip): u8 [B][Hs][Ws][3] -> f32 [B][3][out][out], Pillow-LANCZOS + 1/255 + (x-0.5)/0.5
size_t preprocess_scratch_bytes(int B, int Hs, int Ws, int out_size);
void preprocess_pages(const uint8_t* pages, int B, int Hs, int Ws, int out_size, float* pixel_values, void* scratch, mgStream_t stream);


}  // namespace mg
