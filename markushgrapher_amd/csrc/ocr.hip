// ChemicalOCR stage: the C ABI `mg_ocr_*` of include/mgrapher.h.  What the reference does with
//   processor, model = AutoProcessor / AutoModelForVision2Seq.from_pretrained(model_path)        (markushgrapher/ocr/chemical_ocr.py:76-84)
//   generated_ids = model.generate(**inputs, max_new_tokens=4096, do_sample=False)               (chemical_ocr.py:366-392)
// i.e. stock Idefics3ForConditionalGeneration: SigLIP-style vision tower -> pixel shuffle + projection -> the image tokens of the
// prompt -> Llama-style text model, greedy search with a KV cache.  Host code only orchestrates launches; the contractions run on
// the GEMM / attention kernels of the main path, the glue kernels are in k_ocr.hip.
// Round-2 form: every operation its own launch, fp32 intermediates between GEMM and activation, eager decode steps.
#include "mg_kernels.h"
#include "mg_ocr.h"
#include "mg_swin.h"
#include "../../include/mgrapher.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace mg;

namespace {

int failf(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return fail_msg(code, buf);
}
size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
int round_up(int x, int a) { return (x + a - 1) / a * a; }
int check(const char* what) {
    const int e = mg_peek_error();
    if (e != 0) {
        const MgErrSite site = mg_err_site();
        mg_err_site() = MgErrSite{0, nullptr};
        return failf(MG_E_HIP, "%s: HIP error %d (%s)%s%s", what, e, mg_error_string(e), site.what ? ", first failing call: " : "", site.what ? site.what : "");
    }
    return MG_OK;
}
GemmArgs ga(const uint16_t* X, const uint16_t* W, int M, int N, int K) {
    GemmArgs a{};
    a.X = X; a.W = W; a.M = M; a.N = N; a.K = K;
    return a;
}
struct Raw { size_t off; std::vector<int64_t> shape; size_t n; bool loaded = false; };
struct VLayer { size_t wqkv, wo, fc1, fc2; };
struct TLayer { size_t wqkv, wo, wgu, wd; };
struct Carver {
    char* base;
    size_t off = 0;
    template <typename T> T* take(size_t n) {
        off = align_up(off, 256);
        T* p = base ? (T*)(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

}  // namespace

struct mg_ocr_model {
    mg_ocr_config c;
    int P, g, P_cap, T_img, vka, kvd, qkvn;
    char* arena = nullptr;
    size_t arena_bytes = 0;
    std::map<std::string, Raw> raw;          // HF key -> fp32 copy in the arena
    std::vector<VLayer> vl;
    std::vector<TLayer> tl;
    // decode step: [down_proj of layer l-1 | QKV of layer l] in one launch - the QKV part reads [bf16(h) | y] against the product weight
    // [Wqkv G | Wqkv G Wd] (G = the input_layernorm gain), so it does not wait for the residual update next to it (as the main path's
    // pair projections, engine.hip)
    std::vector<size_t> wqkv2;     // per layer l >= 1; [0] unused
    size_t fin_a = 0, fin_c = 0;
    size_t patch_w, pos_emb, conn, tok_emb, lm_head, zero_tab, rope_cs;
    static constexpr int MAX_POS = 8192;     // positions of the rotation table (prompt + new tokens)
    bool finalized = false;
    mutable std::recursive_mutex call_mu;      // one call at a time per execution context (mg_ocr_clone gives further contexts)
    // one decode step (30 layers x 9 launches + lm_head + selection) captured as a HIP graph whose kernels read the position from the
    // device step counter; replayed while the call's buffers and sizes match (MG_OCR_GRAPH=0: eager launches, same kernels)
    int use_graph = 1;
    bool graph_active = false;
    struct Key { const void *ws, *out, *stream; int B, n_img, L, max_new; bool operator==(const Key& o) const { return ws == o.ws && out == o.out && stream == o.stream && B == o.B && n_img == o.n_img && L == o.L && max_new == o.max_new; } } gkey{};   // n_img: the text-side buffers are carved behind the vision buffers
    bool gvalid = false;
    struct SKey { const void *ws, *out, *out_len, *stream; int N, slots, L, max_new, chunk, n_img, ragged; bool operator==(const SKey& o) const { return ws == o.ws && out == o.out && out_len == o.out_len && stream == o.stream && N == o.N && slots == o.slots && L == o.L && max_new == o.max_new && chunk == o.chunk && n_img == o.n_img && ragged == o.ragged; } } skey{};   // chunk, n_img: the decode rows, K/V pages and slot table are carved behind the prefill region they size
    bool svalid = false;
#ifndef MG_EMU
    hipGraphExec_t sexec = nullptr;
    hipGraphExec_t gexec = nullptr;
    hipStream_t own_stream = nullptr;
    hipEvent_t fork_ev = nullptr;
    void greset() { if (gexec) (void)hipGraphExecDestroy(gexec); gexec = nullptr; gvalid = false; }
    void sreset() { if (sexec) (void)hipGraphExecDestroy(sexec); sexec = nullptr; svalid = false; }
    ~mg_ocr_model() { greset(); sreset(); if (own_stream) (void)hipStreamDestroy(own_stream); if (fork_ev) (void)hipEventDestroy(fork_ev); }
#else
    void greset() { gvalid = false; }
    void sreset() { svalid = false; }
#endif
    template <typename T> T* at(size_t off) const { return (T*)(arena + off); }
    const float* rawp(const std::string& k) const { return (const float*)(arena + raw.at(k).off); }
};

namespace {

struct Ws {
    // vision
    uint16_t *xim, *vx, *vq, *vk, *vvt, *vctx, *vy, *xs;
    float *patch, *vh, *vht, *vtmp, *vout, *feats;
    // text
    float *h, *ht, *qkv, *gu, *logits, *rs_a, *rs_b;
    uint16_t* xw;            // decode step: packed [rows][t_hidden + t_inter] = [bf16(h) | SwiGLU output]
    uint16_t *x, *q, *k, *vt, *ctx, *y, *xc, *Kc, *Vc;
    uint8_t *kmask, *vmask;
    int *last_rows, *all_rows, *unfinished, *counters;
    int64_t* next_ids;
    size_t total, kv_layer;
};

void carve(const mg_ocr_model* m, char* base, int B, int n_img, int L, int max_new, bool full_logits, Ws* w) {
    const mg_ocr_config& c = m->c;
    Carver cv{base};
    const size_t N = (size_t)B * n_img, MV = N * m->P_cap, vh = c.v_hidden, vi = c.v_inter, td = c.t_hidden, ti = c.t_inter;
    const int T_cap = round_up(L, 64), H = c.t_heads;
    const size_t MT = (size_t)B * T_cap;
    const int cap = round_up(L + max_new, 64);
    w->xim = cv.take<uint16_t>(pk_elems((int)(N * m->P), 3 * c.patch_size * c.patch_size));
    w->patch = cv.take<float>(N * m->P * vh);
    w->vh = cv.take<float>(MV * vh);
    w->vht = cv.take<float>(MV * vh);          // the tower's residual stream in the tiled layout (second form)
    w->vx = cv.take<uint16_t>(pk_elems((int)MV, m->vka));
    w->vq = cv.take<uint16_t>(MV * vh); w->vk = cv.take<uint16_t>(MV * vh); w->vvt = cv.take<uint16_t>(MV * vh);
    w->vctx = cv.take<uint16_t>(pk_elems((int)MV, (int)vh));
    w->vtmp = cv.take<float>(MV * vi);
    w->vy = cv.take<uint16_t>(pk_elems((int)MV, (int)vi));
    w->vout = cv.take<float>(MV * vh);
    w->xs = cv.take<uint16_t>(pk_elems((int)(N * m->T_img), (int)(vh * c.scale_factor * c.scale_factor)));
    w->feats = cv.take<float>(N * m->T_img * td);
    w->h = cv.take<float>(MT * td);
    w->ht = cv.take<float>(MT * td);           // the prefill's residual stream in the tiled layout (second form)
    w->x = cv.take<uint16_t>(pk_elems((int)MT, (int)td));
    w->qkv = cv.take<float>(MT * m->qkvn);
    w->q = cv.take<uint16_t>(MT * H * 64); w->k = cv.take<uint16_t>(MT * H * 64); w->vt = cv.take<uint16_t>(MT * H * 64);
    w->ctx = cv.take<uint16_t>(pk_elems((int)MT, H * 64));
    w->gu = cv.take<float>(MT * 2 * ti);
    w->y = cv.take<uint16_t>(pk_elems((int)MT, (int)ti));
    w->xc = cv.take<uint16_t>(pk_elems(full_logits ? (int)(B * L) : round_up(B, 32), (int)td));
    w->logits = cv.take<float>((size_t)round_up(B, 32) * c.vocab);
    w->xw = cv.take<uint16_t>(pk_elems(round_up(B, 32), (int)(td + ti)));
    w->rs_a = cv.take<float>((size_t)round_up(B, 32) * (td / 8)); w->rs_b = cv.take<float>((size_t)round_up(B, 32) * (td / 8));
    w->kv_layer = (size_t)B * c.t_kv_heads * cap * 64;      // decode caches hold the key/value heads once (grouped-query attention)
    w->Kc = cv.take<uint16_t>(w->kv_layer * c.t_layers);
    w->Vc = cv.take<uint16_t>(w->kv_layer * c.t_layers);
    w->kmask = cv.take<uint8_t>(MT);
    w->vmask = cv.take<uint8_t>(MV);
    w->last_rows = cv.take<int>(MT); w->all_rows = cv.take<int>(MT);
    w->unfinished = cv.take<int>(round_up(B, 32)); w->counters = cv.take<int>(16);
    w->next_ids = cv.take<int64_t>(round_up(B, 32));
    w->total = align_up(cv.off, 256);
}

int check_args(const mg_ocr_model* m, int B, int n_img, int L, const char* who) {
    // (every compute entry starts here) whatever is pending in the runtime's per-thread last-error slot was not caused by this call
    (void)mg_peek_error();
    mg_err_site() = MgErrSite{0, nullptr};
    if (!m) return failf(MG_E_ARG, "%s: null model", who);
    if (!m->finalized) return failf(MG_E_STATE, "%s: mg_ocr_finalize has not run", who);
    if (B < 1 || B > 256 || n_img < 0 || L < 1 || L > 2048) return failf(MG_E_SHAPE, "%s: B=%d n_img=%d L=%d out of range", who, B, n_img, L);
    return MG_OK;
}

// vision tower + connector: pixel_values [N][3][I][I] -> w.feats [N*T_img][t_hidden]
void image_features(const mg_ocr_model* m, const Ws& w, const float* pix, const int* patch_pos, const uint8_t* patch_mask, int N, mgStream_t st) {
    const mg_ocr_config& c = m->c;
    const int vh = c.v_hidden, vi = c.v_inter, P = m->P, Pc = m->P_cap, MV = N * Pc, H = c.v_heads, kp = 3 * c.patch_size * c.patch_size;
    const std::string v = "model.vision_model.";
    im2col_pack(pix, w.xim, N, 3, c.image_size, c.patch_size, st);
    GemmArgs pe = ga(w.xim, m->at<uint16_t>(m->patch_w), N * P, vh, kp);
    pe.out_f32 = w.patch; pe.ldo = vh; pe.bias = m->rawp(v + "embeddings.patch_embedding.bias");
    gemm(pe, EPI_F32_STORE, st);
    const bool masked = patch_mask != nullptr || P != Pc;
    ocr_add_pos(w.patch, m->at<uint16_t>(m->pos_emb), patch_pos, patch_mask, masked ? w.vmask : nullptr, w.vh, N, P, Pc, vh, st);
    // Second form (round 5): the residual stream in the main encoder's TILED fp32 layout - the two residual projections of a layer run on
    // the batched 16-byte residual epilogue (EPI_RESID_NORM without gain; the row-major read-modify-write epilogue was 273 us per launch
    // at 32 pages), LayerNorm on the tile-wise kernel of the OCSR branch (512-byte runs; constant-one column and the bias hand-off as
    // before).  Widths that kernel is not instantiated for keep the first form.  MG_OCR_TILED=0: first form (A/B runs; same operations).
    static int tiled_env = -1;
    if (tiled_env < 0) { const char* e = getenv("MG_OCR_TILED"); tiled_env = e ? atoi(e) : 1; }
    const bool tiled = tiled_env && (MV % 32) == 0 && (vh == 64 || vh == 128 || vh == 256 || vh == 512 || vh == 768 || vh == 1024);
    auto layer_norm = [&](bool first, const float* wt, const float* bs, const float* add_bias, uint16_t* x_pk, float* out_f32, int Kaug) {
        if (!tiled) { ocr_layernorm_pack(w.vh, wt, bs, add_bias, x_pk, out_f32, MV, vh, Kaug, c.v_eps, st); return; }
        SwinLnArgs n{};
        n.h_in = first ? w.vh : w.vht; n.in_tiled = first ? 0 : 1; n.h_out = (first || add_bias) ? w.vht : nullptr;
        n.w = wt; n.b = bs; n.add_bias = add_bias; n.x_pk = x_pk; n.out_f32 = out_f32; n.M = MV; n.C = vh; n.eps = c.v_eps; n.kaug = Kaug;
        swin_layernorm(n, st);
    };
    for (int i = 0; i < c.v_layers; ++i) {
        const std::string p = v + "encoder.layers." + std::to_string(i) + ".";
        const VLayer& l = m->vl[i];
        layer_norm(i == 0, m->rawp(p + "layer_norm1.weight"), m->rawp(p + "layer_norm1.bias"), m->rawp(p + "self_attn.out_proj.bias"), w.vx, nullptr, m->vka);
        GemmArgs a = ga(w.vx, m->at<uint16_t>(l.wqkv), MV, 3 * vh, m->vka);
        a.heads.ptr[0] = w.vq; a.heads.ptr[1] = w.vk; a.heads.ptr[2] = w.vvt;
        a.heads.fmt[0] = HF_PK_ROWS; a.heads.fmt[1] = HF_PK_ROWS; a.heads.fmt[2] = HF_PK_T;
        a.heads.inner = vh; a.heads.H = H; a.heads.S_in = Pc; a.heads.S_cap = Pc;
        gemm(a, EPI_HEADS, st);
        AttnArgs t{};
        t.Q = w.vq; t.K = w.vk; t.Vt = w.vvt; t.ctx = w.vctx; t.B = N; t.H = H; t.Sq = P; t.Sk = P; t.Sq_cap = Pc; t.Sk_cap = Pc;
        t.mode = ATT_CROSS; t.kmask = patch_mask ? w.vmask : nullptr;      // padded frames: masked patches are not attended as keys
        attention(t, st);
        GemmArgs o = ga(w.vctx, m->at<uint16_t>(l.wo), MV, vh, vh);
        if (tiled) { o.out_f32 = w.vht; gemm(o, EPI_RESID_NORM, st); }
        else { o.out_f32 = w.vh; o.ldo = vh; gemm(o, EPI_F32_RESID, st); }
        // fc1: K = v_hidden, its bias in the epilogue (the QKV projection keeps its bias in the constant-one column of K: the per-head epilogue has no
        // bias slot) - K = 768 is a multiple of the ping-pong kernel's 128 where 832 is not: 218 -> us per launch at 32 pages
        layer_norm(false, m->rawp(p + "layer_norm2.weight"), m->rawp(p + "layer_norm2.bias"), m->rawp(p + "mlp.fc2.bias"), w.vx, nullptr, vh);
        GemmArgs f1 = ga(w.vx, m->at<uint16_t>(l.fc1), MV, vi, vh);
        f1.bias = m->rawp(p + "mlp.fc1.bias");
        if (gemm_has_gelu_epilogue(MV, vi)) {          // GELU in the GEMM epilogue: no fp32 round trip of [rows][v_inter]
            f1.out_pk = w.vy;
            gemm(f1, EPI_PK_GELU, st);
        } else {                                       // small shapes (test fixtures): fp32 store (+ bias) + activation kernel
            f1.out_f32 = w.vtmp; f1.ldo = vi;
            gemm(f1, EPI_F32_STORE, st);
            ocr_gelu_pack(w.vtmp, w.vy, MV, vi, vi, st);
        }
        GemmArgs f2 = ga(w.vy, m->at<uint16_t>(l.fc2), MV, vh, vi);
        if (tiled) { f2.out_f32 = w.vht; gemm(f2, EPI_RESID_NORM, st); }
        else { f2.out_f32 = w.vh; f2.ldo = vh; gemm(f2, EPI_F32_RESID, st); }
    }
    layer_norm(c.v_layers == 0, m->rawp(v + "post_layernorm.weight"), m->rawp(v + "post_layernorm.bias"), nullptr, nullptr, w.vout, vh);
    const int sf = c.scale_factor, F = vh * sf * sf;
    ocr_pixel_shuffle_pack(w.vout, w.xs, N, m->g, Pc, vh, sf, st);
    GemmArgs cn = ga(w.xs, m->at<uint16_t>(m->conn), N * m->T_img, c.t_hidden, F);
    cn.out_f32 = w.feats; cn.ldo = c.t_hidden;
    gemm(cn, EPI_F32_STORE, st);
}

// text model over the whole prompt (teacher-forced / prefill): leaves the final hidden states in w.h (before the last norm)
// and the rotated keys / values of every layer in the decode caches
// lens (nullable, device [B]): prompts of different lengths, left-aligned in their rows (row b: lens[b] tokens, then padding): the padding is
// behind every real token (causal attention never reads it) and its keys are masked; the row's last position is lens[b] - 1
void prefill(const mg_ocr_model* m, const Ws& w, const int64_t* ids, const float* feats, int B, int n_img, int L, int cap, mgStream_t st, const int* lens = nullptr) {
    const mg_ocr_config& c = m->c;
    const int td = c.t_hidden, ti = c.t_inter, H = c.t_heads, KV = c.t_kv_heads, T_cap = round_up(L, 64), MT = B * T_cap;
    ocr_row_maps(w.last_rows, w.all_rows, w.kmask, B, L, T_cap, st, lens);
    ocr_merge_embed(ids, m->at<uint16_t>(m->tok_emb), feats, w.h, B, L, T_cap, td, c.vocab, c.image_token_id, n_img * m->T_img, w.counters + 3, st);
    // Second form (round 5): the residual stream of the prefill in the TILED fp32 layout (w.ht) - o_proj / down_proj on the batched residual
    // epilogue instead of the row-major read-modify-write one, RMSNorm from 512-byte runs; the row-major copy w.h (what the callers read the
    // last positions from) is restored at the end.  MG_OCR_TILED=0: first form.
    static int tiled_env = -1;
    if (tiled_env < 0) { const char* e = getenv("MG_OCR_TILED"); tiled_env = e ? atoi(e) : 1; }
    const bool tiled = tiled_env && (td % 32) == 0;
    if (tiled) ocr_tile_f32(w.h, w.ht, MT, td, 1, st);
    auto norm = [&](const float* gain) {
        if (tiled) rmsnorm_pack_tiled(w.ht, gain, w.x, nullptr, MT, td, c.rms_eps, st);
        else rmsnorm_pack(w.h, gain, w.x, nullptr, MT, td, c.rms_eps, 1.0f, st);
    };
    auto resid = [&](GemmArgs& g) {
        if (tiled) { g.out_f32 = w.ht; gemm(g, EPI_RESID_NORM, st); }
        else { g.out_f32 = w.h; g.ldo = td; gemm(g, EPI_F32_RESID, st); }
    };
    for (int i = 0; i < c.t_layers; ++i) {
        const std::string p = "model.text_model.layers." + std::to_string(i) + ".";
        const TLayer& l = m->tl[i];
        norm(m->rawp(p + "input_layernorm.weight"));
        GemmArgs a = ga(w.x, m->at<uint16_t>(l.wqkv), MT, m->qkvn, td);
        a.out_f32 = w.qkv; a.ldo = m->qkvn;
        gemm(a, EPI_F32_STORE, st);
        ocr_rope_heads(w.qkv, B, L, T_cap, H, KV, c.rope_theta, w.q, w.k, w.vt, w.Kc + (size_t)i * w.kv_layer, w.Vc + (size_t)i * w.kv_layer, cap, st);
        AttnArgs t{};
        t.Q = w.q; t.K = w.k; t.Vt = w.vt; t.ctx = w.ctx; t.B = B; t.H = H; t.Sq = L; t.Sk = L; t.Sq_cap = T_cap; t.Sk_cap = T_cap;
        t.mode = ATT_DEC_SELF; t.kmask = w.kmask; t.tab1 = m->at<float>(m->zero_tab); t.tab1_len = 1;
        attention(t, st);
        GemmArgs o = ga(w.ctx, m->at<uint16_t>(l.wo), MT, td, H * 64);
        resid(o);
        norm(m->rawp(p + "post_attention_layernorm.weight"));
        GemmArgs gu = ga(w.x, m->at<uint16_t>(l.wgu), MT, 2 * ti, td);
        gu.out_f32 = w.gu; gu.ldo = 2 * ti;
        gemm(gu, EPI_F32_STORE, st);
        ocr_silu_mul_pack(w.gu, w.y, MT, ti, st);
        GemmArgs dn = ga(w.y, m->at<uint16_t>(l.wd), MT, td, ti);
        resid(dn);
    }
    if (tiled) ocr_tile_f32(w.ht, w.h, MT, td, 0, st);
}

// One decode step for B rows: token ids in w.next_ids, position `pos`; logits -> w.logits.
// pos_dev != null: position = *pos_dev + pos (graph replay: pos_dev = the step counter, pos = prompt length - 1).
// Runs on the decode path's deferred-RMSNorm kernels: the residual projections
// (o_proj, down_proj) leave bf16(h * gain_next) un-normalised plus per-row partial sums of h^2, and the consumers of the
// projections that read it (rotary / cache kernel, SiLU kernel, lm_head) apply r(row) = rsqrt(mean h^2 + eps) themselves.
// q != null (queue form, mg_ocr_generate_stream): the B rows are SLOTS - row r decodes page q->pool[r] at its own position
// pos + q->pos[r] and reads / appends to that page's cache; idle slots are skipped by the attention launches.
struct SlotView { const int* pos; const int* pool; const int* live; const int* toff; };      // toff (nullable): per page, its prompt length - L
void decode_step(const mg_ocr_model* m, const Ws& w, int B, int pos, const int* pos_dev, int cap, mgStream_t st, const SlotView* q = nullptr) {
    const mg_ocr_config& c = m->c;
    const int td = c.t_hidden, ti = c.t_inter, H = c.t_heads, KV = c.t_kv_heads, K2 = td + ti;
    const RowScale none{};
    const RowScale rs_a{w.rs_a, td / 8, 1.0f / (float)td, c.rms_eps};      // after a layer's MLP (next input_layernorm / final norm)
    const RowScale rs_b{w.rs_b, td / 8, 1.0f / (float)td, c.rms_eps};      // after the attention (post_attention_layernorm)
    embed_norm_rows(w.next_ids, m->at<uint16_t>(m->tok_emb), w.h, m->rawp("model.text_model.layers.0.input_layernorm.weight"), w.x, nullptr, 0, 0,
                    B, td, c.vocab, w.counters + 3, c.rms_eps, st);
    // One row tile of sequences: the next layer's QKV rides in the down_proj launch (product weights).  More rows: the pair kernel has
    // no row-tile split form, the separate launches (which split) are faster (measured at 128 pages: 1.40 vs 1.34 ms per step).
    const bool pair = B <= 32;
    for (int i = 0; i < c.t_layers; ++i) {
        const std::string p = "model.text_model.layers." + std::to_string(i) + ".";
        const TLayer& l = m->tl[i];
        if (i == 0 || !pair) {   // QKV from x: the explicitly normalised embedding (layer 0) or bf16(h gain) un-normalised (rs_a applied by the attention)
            GemmArgs a = ga(w.x, m->at<uint16_t>(l.wqkv), B, m->qkvn, td);
            a.out_f32 = w.qkv; a.ldo = m->qkvn;
            gemm_rows(a, EPI_F32_STORE, st);
        }
        uint16_t* Kc = w.Kc + (size_t)i * w.kv_layer;
        uint16_t* Vc = w.Vc + (size_t)i * w.kv_layer;
        AttnStepArgs s{};         // rotary embedding, cache append and grouped-query attention over [0, pos] in one launch
        s.Kc = Kc; s.Vc = Vc; s.Kc_w = Kc; s.Vc_w = Vc; s.ctx = w.ctx; s.rows = B; s.H = KV; s.group = H / KV; s.cap = cap; s.n_keys = pos + 1; s.t = pos;
        s.t_dev = pos_dev; s.t_off = pos;
        if (q) { s.pos_rows = q->pos; s.kv_owner = q->pool; s.live = q->live; s.t_off_rows = q->toff; }
        s.rope.qkv = w.qkv; s.rope.ld = m->qkvn; s.rope.cs = m->at<float>(m->rope_cs); s.rope.rs = i == 0 ? none : rs_a;
        s.rope.qscale = 0.125f;
        attention_step(s, st);
        ResidArgs o{};            // h += Wo ctx;  x = bf16(h gain_post) un-normalised;  window[:, :td] = bf16(h);  partials -> rs_b
        o.X = w.ctx; o.W = m->at<uint16_t>(l.wo); o.h = w.h; o.gain = m->rawp(p + "post_attention_layernorm.weight"); o.gscale = 1.0f;
        o.x_pk = w.x; o.x2_pk = w.xw; o.x2_ld = K2; o.x2_col0 = 0; o.part = w.rs_b; o.M = B; o.N = td; o.K = H * 64;
        gemm_rows_resid(o, st);
        GemmArgs gu = ga(w.x, m->at<uint16_t>(l.wgu), B, 2 * ti, td);
        gu.out_pk = w.xw; gu.out_ld = K2; gu.out_col0 = td; gu.rs = rs_b;
        gemm_rows(gu, EPI_PK_SWIGLU, st);           // window[:, td:] = silu(r gate) * (r up)
        ResidArgs d{};            // h += Wd y;  partials -> rs_a
        d.X = w.xw; d.x_kts = K2 >> 4; d.x_k0 = td >> 4; d.W = m->at<uint16_t>(l.wd); d.h = w.h; d.gscale = 1.0f;
        d.part = w.rs_a; d.M = B; d.N = td; d.K = ti;
        if (pair && i + 1 < c.t_layers) {
            // ... side by side with the next layer's QKV = [Wqkv G | Wqkv G Wd] [bf16(h) ; y]  (un-normalised: the attention applies rs_a)
            GemmArgs q = ga(w.xw, m->at<uint16_t>(m->wqkv2[i + 1]), B, m->qkvn, K2);
            q.out_f32 = w.qkv; q.ldo = m->qkvn;
            gemm_rows_pair(d, q, EPI_F32_STORE, st);
        } else {
            d.gain = m->rawp(i + 1 < c.t_layers ? "model.text_model.layers." + std::to_string(i + 1) + ".input_layernorm.weight" : std::string("model.text_model.norm.weight"));
            d.x_pk = w.x;
            gemm_rows_resid(d, st);
        }
    }
    gemm_rows_splitk(w.x, m->at<uint16_t>(m->lm_head), w.logits, B, c.vocab, td, c.vocab, 0, 1, rs_a, st);
}

__global__ __launch_bounds__(256) void copy_f32_kernel(const float* src, float* dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

extern "C" {

int mg_ocr_create(const mg_ocr_config* cfg, mg_ocr_model** out) {
    if (!cfg || !out) return failf(MG_E_ARG, "mg_ocr_create: null argument");
    const mg_ocr_config& c = *cfg;
    if (c.v_hidden % 64 || c.v_hidden != c.v_heads * 64 || c.t_hidden != c.t_heads * 64 || c.t_heads % c.t_kv_heads || c.v_inter % 64 ||
        c.t_inter % 64 || c.t_hidden % 64 || c.vocab % 32 || c.image_size % c.patch_size || (3 * c.patch_size * c.patch_size) % 64)
        return failf(MG_E_SHAPE, "mg_ocr_create: unsupported geometry (head dim must be 64, widths multiples of 64, vocab of 32)");
    { const int rep = c.t_heads / c.t_kv_heads;
      if (!(rep == 1 || rep == 2 || rep == 3 || rep == 4 || rep == 6 || rep == 8))
          return failf(MG_E_SHAPE, "mg_ocr_create: %d query heads per key/value head (supported: 1, 2, 3, 4, 6, 8)", rep); }
    if (c.n_eos_extra < 0 || c.n_eos_extra > 3) return failf(MG_E_SHAPE, "mg_ocr_create: at most 4 stop tokens (eos_token_id + 3 extra), got %d extra", c.n_eos_extra);
    const int g = c.image_size / c.patch_size;
    if (g % c.scale_factor) return failf(MG_E_SHAPE, "mg_ocr_create: patch grid %d not divisible by scale_factor %d", g, c.scale_factor);
    mg_ocr_model* m = new mg_ocr_model();
    m->c = c; m->g = g; m->P = g * g; m->P_cap = round_up(m->P, 64); m->T_img = m->P / (c.scale_factor * c.scale_factor);
    m->vka = c.v_hidden + 64; m->kvd = c.t_kv_heads * 64; m->qkvn = (c.t_heads + 2 * c.t_kv_heads) * 64;
    // raw fp32 copies, keyed by the HF names of stock Idefics3ForConditionalGeneration
    size_t off = 0;
    auto add = [&](const std::string& k, std::vector<int64_t> shape) {
        size_t n = 1;
        for (int64_t s : shape) n *= (size_t)s;
        off = align_up(off, 256);
        m->raw[k] = Raw{off, shape, n};
        off += n * sizeof(float);
    };
    const int64_t vh = c.v_hidden, vi = c.v_inter, td = c.t_hidden, ti = c.t_inter;
    const std::string v = "model.vision_model.";
    add(v + "embeddings.patch_embedding.weight", {vh, 3, c.patch_size, c.patch_size});
    add(v + "embeddings.patch_embedding.bias", {vh});
    add(v + "embeddings.position_embedding.weight", {m->P, vh});
    for (int i = 0; i < c.v_layers; ++i) {
        const std::string p = v + "encoder.layers." + std::to_string(i) + ".";
        for (const char* n : {"k_proj", "v_proj", "q_proj", "out_proj"}) {
            add(p + "self_attn." + n + ".weight", {vh, vh});
            add(p + "self_attn." + n + ".bias", {vh});
        }
        add(p + "layer_norm1.weight", {vh}); add(p + "layer_norm1.bias", {vh});
        add(p + "mlp.fc1.weight", {vi, vh}); add(p + "mlp.fc1.bias", {vi});
        add(p + "mlp.fc2.weight", {vh, vi}); add(p + "mlp.fc2.bias", {vh});
        add(p + "layer_norm2.weight", {vh}); add(p + "layer_norm2.bias", {vh});
    }
    add(v + "post_layernorm.weight", {vh}); add(v + "post_layernorm.bias", {vh});
    add("model.connector.modality_projection.proj.weight", {td, vh * c.scale_factor * c.scale_factor});
    add("model.text_model.embed_tokens.weight", {c.vocab, td});
    for (int i = 0; i < c.t_layers; ++i) {
        const std::string p = "model.text_model.layers." + std::to_string(i) + ".";
        add(p + "self_attn.q_proj.weight", {td, td});
        add(p + "self_attn.k_proj.weight", {m->kvd, td});
        add(p + "self_attn.v_proj.weight", {m->kvd, td});
        add(p + "self_attn.o_proj.weight", {td, td});
        add(p + "mlp.gate_proj.weight", {ti, td});
        add(p + "mlp.up_proj.weight", {ti, td});
        add(p + "mlp.down_proj.weight", {td, ti});
        add(p + "input_layernorm.weight", {td});
        add(p + "post_attention_layernorm.weight", {td});
    }
    add("model.text_model.norm.weight", {td});
    if (!c.tie_word_embeddings) add("lm_head.weight", {c.vocab, td});
    // packed bf16 operands built by mg_ocr_finalize
    auto pk = [&](int N, int K) { off = align_up(off, 256); const size_t o = off; off += pk_elems(round_up(N, 32), K) * 2; return o; };
    m->patch_w = pk((int)vh, 3 * c.patch_size * c.patch_size);
    off = align_up(off, 256); m->pos_emb = off; off += (size_t)m->P * vh * 2;
    for (int i = 0; i < c.v_layers; ++i) m->vl.push_back(VLayer{pk(3 * (int)vh, m->vka), pk((int)vh, (int)vh), pk((int)vi, m->vka), pk((int)vh, (int)vi)});
    m->conn = pk((int)td, (int)(vh * c.scale_factor * c.scale_factor));
    off = align_up(off, 256); m->tok_emb = off; off += (size_t)c.vocab * td * 2;
    for (int i = 0; i < c.t_layers; ++i) m->tl.push_back(TLayer{pk(m->qkvn, (int)td), pk((int)td, (int)td), pk(2 * (int)ti, (int)td), pk((int)td, (int)ti)});
    m->lm_head = pk(c.vocab, (int)td);
    m->wqkv2.assign(c.t_layers, 0);
    for (int i = 1; i < c.t_layers; ++i) m->wqkv2[i] = pk(m->qkvn, (int)(td + ti));
    off = align_up(off, 256); m->fin_a = off; off += (size_t)m->qkvn * td * sizeof(float);
    off = align_up(off, 256); m->fin_c = off; off += (size_t)m->qkvn * (td + ti) * sizeof(float);
    off = align_up(off, 256); m->zero_tab = off; off += 64 * sizeof(float);
    off = align_up(off, 256); m->rope_cs = off; off += (size_t)mg_ocr_model::MAX_POS * 64 * sizeof(float);
    m->arena_bytes = align_up(off, 256);
    { const char* e = getenv("MG_OCR_GRAPH"); if (e && e[0] == '0') m->use_graph = 0; }
    { const char* e = getenv("AMD_DIRECT_DISPATCH"); if (e && e[0] == '0') m->use_graph = 0; }     // (see mg_create)
    *out = m;
    return MG_OK;
}

void mg_ocr_destroy(mg_ocr_model* m) { delete m; }

// A further execution context on a finalized model's weights (as mg_clone, engine.hip): same arena, own captured graphs.
int mg_ocr_clone(const mg_ocr_model* src, mg_ocr_model** out) {
    if (!src || !out) return fail_msg(MG_E_ARG, "mg_ocr_clone: null argument");
    if (!src->arena || !src->finalized) return fail_msg(MG_E_STATE, "mg_ocr_clone: the source model has no finalized weights");
    mg_ocr_model* m = new mg_ocr_model();
    m->c = src->c;
    m->P = src->P; m->g = src->g; m->P_cap = src->P_cap; m->T_img = src->T_img; m->vka = src->vka; m->kvd = src->kvd; m->qkvn = src->qkvn;
    m->arena = src->arena; m->arena_bytes = src->arena_bytes;
    m->raw = src->raw; m->vl = src->vl; m->tl = src->tl; m->wqkv2 = src->wqkv2;
    m->fin_a = src->fin_a; m->fin_c = src->fin_c;
    m->patch_w = src->patch_w; m->pos_emb = src->pos_emb; m->conn = src->conn; m->tok_emb = src->tok_emb; m->lm_head = src->lm_head;
    m->zero_tab = src->zero_tab; m->rope_cs = src->rope_cs;
    m->finalized = true; m->use_graph = src->use_graph;
    *out = m;
    return MG_OK;
}
size_t mg_ocr_weights_bytes(const mg_ocr_model* m) { return m ? m->arena_bytes : 0; }
int mg_ocr_bind_weights(mg_ocr_model* m, void* arena) {
    if (!m || !arena) return failf(MG_E_ARG, "mg_ocr_bind_weights: null argument");
    m->arena = (char*)arena;
    m->finalized = false;
    return MG_OK;
}

int mg_ocr_load_tensor(mg_ocr_model* m, void* stream, const char* hf_key, const void* src, int src_is_bf16, const int64_t* shape, int ndim) {
    if (!m || !hf_key || !src) return failf(MG_E_ARG, "mg_ocr_load_tensor: null argument");
    if (!m->arena) return failf(MG_E_STATE, "mg_ocr_load_tensor: no weights arena bound");
    auto it = m->raw.find(hf_key);
    if (it == m->raw.end()) return failf(MG_E_KEY, "mg_ocr_load_tensor: unknown key '%s'", hf_key);
    Raw& r = it->second;
    if ((size_t)ndim != r.shape.size()) return failf(MG_E_SHAPE, "mg_ocr_load_tensor: '%s' has %d dims, expected %zu", hf_key, ndim, r.shape.size());
    for (int i = 0; i < ndim; ++i)
        if (shape[i] != r.shape[i]) return failf(MG_E_SHAPE, "mg_ocr_load_tensor: '%s' dim %d is %lld, expected %lld", hf_key, i, (long long)shape[i], (long long)r.shape[i]);
    convert_to_f32(src, src_is_bf16, (float*)(m->arena + r.off), r.n, (mgStream_t)stream);
    r.loaded = true;
    m->finalized = false;
    return check(hf_key);
}

int mg_ocr_finalize(mg_ocr_model* m, void* stream) {
    if (!m || !m->arena) return failf(MG_E_STATE, "mg_ocr_finalize: no model / arena");
    for (auto& kv : m->raw)
        if (!kv.second.loaded) return failf(MG_E_STATE, "mg_ocr_finalize: tensor '%s' was not loaded", kv.first.c_str());
    mgStream_t st = (mgStream_t)stream;
    const mg_ocr_config& c = m->c;
    const int vh = c.v_hidden, vi = c.v_inter, td = c.t_hidden, ti = c.t_inter;
    const std::string v = "model.vision_model.";
    auto pack = [&](const std::string& k, size_t dst, int row0, int N, int K, int Kaug, const std::string& bias, float scale, int Nfill, int rstride = 1) {
        ocr_pack_aug(m->rawp(k), bias.empty() ? nullptr : m->rawp(bias), scale, m->at<uint16_t>(dst), row0, N, K, Kaug, Nfill, rstride, st);
    };
    pack(v + "embeddings.patch_embedding.weight", m->patch_w, 0, vh, 3 * c.patch_size * c.patch_size, 3 * c.patch_size * c.patch_size, "", 1.f, round_up(vh, 32));
    convert_to_bf16(m->rawp(v + "embeddings.position_embedding.weight"), 0, m->at<uint16_t>(m->pos_emb), (size_t)m->P * vh, st);
    for (int i = 0; i < c.v_layers; ++i) {
        const std::string p = v + "encoder.layers." + std::to_string(i) + ".";
        const VLayer& l = m->vl[i];
        // q | k | v with their biases in the constant-one column; the softmax scale 64^-0.5 = 2^-3 folded into q (exact in bf16)
        pack(p + "self_attn.q_proj.weight", l.wqkv, 0, vh, vh, m->vka, p + "self_attn.q_proj.bias", 0.125f, vh);
        pack(p + "self_attn.k_proj.weight", l.wqkv, vh, vh, vh, m->vka, p + "self_attn.k_proj.bias", 1.f, vh);
        pack(p + "self_attn.v_proj.weight", l.wqkv, 2 * vh, vh, vh, m->vka, p + "self_attn.v_proj.bias", 1.f, round_up(3 * vh, 32) - 2 * vh);
        pack(p + "self_attn.out_proj.weight", l.wo, 0, vh, vh, vh, "", 1.f, round_up(vh, 32));
        pack(p + "mlp.fc1.weight", l.fc1, 0, vi, vh, vh, "", 1.f, round_up(vi, 32));              // (bias: epilogue of the projection)
        pack(p + "mlp.fc2.weight", l.fc2, 0, vh, vi, vi, "", 1.f, round_up(vh, 32));
    }
    const int F = vh * c.scale_factor * c.scale_factor;
    pack("model.connector.modality_projection.proj.weight", m->conn, 0, td, F, F, "", 1.f, round_up(td, 32));
    convert_to_bf16(m->rawp("model.text_model.embed_tokens.weight"), 0, m->at<uint16_t>(m->tok_emb), (size_t)c.vocab * td, st);
    for (int i = 0; i < c.t_layers; ++i) {
        const std::string p = "model.text_model.layers." + std::to_string(i) + ".";
        const TLayer& l = m->tl[i];
        pack(p + "self_attn.q_proj.weight", l.wqkv, 0, td, td, td, "", 1.f, td);
        pack(p + "self_attn.k_proj.weight", l.wqkv, td, m->kvd, td, td, "", 1.f, m->kvd);
        pack(p + "self_attn.v_proj.weight", l.wqkv, td + m->kvd, m->kvd, td, td, "", 1.f, round_up(m->qkvn, 32) - td - m->kvd);
        pack(p + "self_attn.o_proj.weight", l.wo, 0, td, td, td, "", 1.f, round_up(td, 32));
        pack(p + "mlp.gate_proj.weight", l.wgu, 0, ti, td, td, "", 1.f, ti, 2);       // gate_j -> row 2j, up_j -> row 2j + 1 (EPI_PK_SWIGLU)
        pack(p + "mlp.up_proj.weight", l.wgu, 1, ti, td, td, "", 1.f, ti, 2);
        pack(p + "mlp.down_proj.weight", l.wd, 0, td, ti, ti, "", 1.f, round_up(td, 32));
    }
    pack(c.tie_word_embeddings ? "model.text_model.embed_tokens.weight" : "lm_head.weight", m->lm_head, 0, c.vocab, td, td, "", 1.f, round_up(c.vocab, 32));
    {   // product weights of the decode step's [down_proj | next QKV] launches, fp32 then one rounding to bf16
        float* A = m->at<float>(m->fin_a);
        float* C = m->at<float>(m->fin_c);
        const int K2 = td + ti;
        for (int i = 1; i < c.t_layers; ++i) {
            const std::string p = "model.text_model.layers." + std::to_string(i) + ".", q = "model.text_model.layers." + std::to_string(i - 1) + ".";
            mg_memcpy_async(A, m->rawp(p + "self_attn.q_proj.weight"), (size_t)td * td * sizeof(float), st);
            mg_memcpy_async(A + (size_t)td * td, m->rawp(p + "self_attn.k_proj.weight"), (size_t)m->kvd * td * sizeof(float), st);
            mg_memcpy_async(A + (size_t)(td + m->kvd) * td, m->rawp(p + "self_attn.v_proj.weight"), (size_t)m->kvd * td * sizeof(float), st);
            const float* gain = m->rawp(p + "input_layernorm.weight");
            scale_cols_f32(A, gain, C, m->qkvn, td, K2, st);
            gemm_f32_scaled(A, gain, m->rawp(q + "mlp.down_proj.weight"), C + td, m->qkvn, td, ti, K2, st);
            pack_weight(C, 0, m->qkvn, K2, m->at<uint16_t>(m->wqkv2[i]), round_up(m->qkvn, 32), st);
        }
    }
    mg_memset_async(m->at<float>(m->zero_tab), 0, 64 * sizeof(float), st);
    ocr_rope_table(m->at<float>(m->rope_cs), mg_ocr_model::MAX_POS, c.rope_theta, st);
    const int rc = check("mg_ocr_finalize");
    if (rc == MG_OK) m->finalized = true;
    return rc;
}

int mg_ocr_workspace_bytes(const mg_ocr_model* m, int B, int n_img, int L, int max_new_tokens, int full_logits, size_t* out_bytes) {
    if (!m || !out_bytes) return failf(MG_E_ARG, "mg_ocr_workspace_bytes: null argument");
    Ws w;
    carve(m, nullptr, B, n_img, L, max_new_tokens, full_logits != 0, &w);
    *out_bytes = w.total;
    return MG_OK;
}

int mg_ocr_image_features(mg_ocr_model* m, void* stream, void* ws, size_t ws_bytes, const float* pixel_values, const int32_t* patch_pos,
                          const uint8_t* patch_mask, int N, float* out) {
    int rc = check_args(m, N, 1, 1, "mg_ocr_image_features");
    if (rc != MG_OK) return rc;
    std::unique_lock<std::recursive_mutex> call_lock;
    if (m) {
        call_lock = std::unique_lock<std::recursive_mutex>(m->call_mu, std::try_to_lock);
        if (!call_lock.owns_lock()) return failf(MG_E_STATE, "mg_ocr_image_features: this execution context is inside another call (one call at a time per context; mg_ocr_clone gives further contexts)");
    }
    Ws w;
    carve(m, (char*)ws, N, 1, 1, 0, false, &w);
    if (!ws || ws_bytes < w.total) return failf(MG_E_WORKSPACE, "mg_ocr_image_features: workspace %zu < %zu bytes", ws_bytes, w.total);
    mgStream_t st = (mgStream_t)stream;
    if ((patch_pos == nullptr) != (patch_mask == nullptr)) return failf(MG_E_ARG, "mg_ocr_image_features: patch_pos and patch_mask go together");
    image_features(m, w, pixel_values, patch_pos, patch_mask, N, st);
    const size_t n = (size_t)N * m->T_img * m->c.t_hidden;
    MG_LAUNCH(copy_f32_kernel, dim3(256), dim3(256), 0, st, (const float*)w.feats, out, n);
    return check("mg_ocr_image_features");
}

int mg_ocr_forward(mg_ocr_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* pixel_values,
                   const int32_t* patch_pos, const uint8_t* patch_mask, int B, int n_img, int L, float* logits) {
    int rc = check_args(m, B, n_img, L, "mg_ocr_forward");
    if (rc != MG_OK) return rc;
    std::unique_lock<std::recursive_mutex> call_lock;
    if (m) {
        call_lock = std::unique_lock<std::recursive_mutex>(m->call_mu, std::try_to_lock);
        if (!call_lock.owns_lock()) return failf(MG_E_STATE, "mg_ocr_forward: this execution context is inside another call (one call at a time per context; mg_ocr_clone gives further contexts)");
    }
    Ws w;
    carve(m, (char*)ws, B, n_img, L, 0, true, &w);
    if (!ws || ws_bytes < w.total) return failf(MG_E_WORKSPACE, "mg_ocr_forward: workspace %zu < %zu bytes", ws_bytes, w.total);
    mgStream_t st = (mgStream_t)stream;
    const mg_ocr_config& c = m->c;
    mg_memset_async(w.counters, 0, 16 * sizeof(int), st);
    if ((patch_pos == nullptr) != (patch_mask == nullptr)) return failf(MG_E_ARG, "mg_ocr_forward: patch_pos and patch_mask go together");
    if (pixel_values && n_img > 0) image_features(m, w, pixel_values, patch_pos, patch_mask, B * n_img, st);
    prefill(m, w, input_ids, (pixel_values && n_img > 0) ? w.feats : nullptr, B, n_img, L, round_up(L, 64), st);
    const int T_cap = round_up(L, 64);
    rmsnorm_pack_rows(w.h, m->rawp("model.text_model.norm.weight"), w.xc, w.all_rows, B * T_cap, c.t_hidden, c.rms_eps, 1.0f, st);
    GemmArgs lg = ga(w.xc, m->at<uint16_t>(m->lm_head), B * L, c.vocab, c.t_hidden);
    lg.out_f32 = logits; lg.ldo = c.vocab;
    gemm(lg, EPI_F32_STORE, st);
    int bad = 0;
    mg_memcpy_async(&bad, w.counters + 3, sizeof(int), st);
    mg_stream_sync(st);
    rc = check("mg_ocr_forward");
    if (rc != MG_OK) return rc;
    if (bad) return failf(MG_E_INPUT, "mg_ocr_forward: %d bad inputs (token id outside the vocabulary, or a sequence whose <image> count differs from n_img * %d)", bad, m->T_img);
    return MG_OK;
}

int mg_ocr_generate(mg_ocr_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* pixel_values,
                    const int32_t* patch_pos, const uint8_t* patch_mask, int B, int n_img, int L, int max_new_tokens, int64_t* out_ids,
                    int* out_cols_host, float* step_logits, int capture_steps) {
    int rc = check_args(m, B, n_img, L, "mg_ocr_generate");
    if (rc != MG_OK) return rc;
    std::unique_lock<std::recursive_mutex> call_lock;
    if (m) {
        call_lock = std::unique_lock<std::recursive_mutex>(m->call_mu, std::try_to_lock);
        if (!call_lock.owns_lock()) return failf(MG_E_STATE, "mg_ocr_generate: this execution context is inside another call (one call at a time per context; mg_ocr_clone gives further contexts)");
    }
    if (max_new_tokens < 1 || !out_ids || !out_cols_host) return failf(MG_E_ARG, "mg_ocr_generate: bad output arguments");
    if (L + max_new_tokens > mg_ocr_model::MAX_POS) return failf(MG_E_SHAPE, "mg_ocr_generate: %d + %d positions exceed %d", L, max_new_tokens, mg_ocr_model::MAX_POS);
    Ws w;
    carve(m, (char*)ws, B, n_img, L, max_new_tokens, false, &w);
    if (!ws || ws_bytes < w.total) return failf(MG_E_WORKSPACE, "mg_ocr_generate: workspace %zu < %zu bytes", ws_bytes, w.total);
    mgStream_t st = (mgStream_t)stream;
#ifndef MG_EMU
    // the legacy null stream cannot be captured: the call then runs on a stream the model owns, ordered after the caller's stream
    // by an event and host-synchronised before returning (as mg_generate does)
    if (m->use_graph == 1 && st == nullptr && !step_logits) {
        if (!m->own_stream && hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess) m->own_stream = nullptr;
        if (!m->fork_ev && hipEventCreateWithFlags(&m->fork_ev, hipEventDisableTiming) != hipSuccess) m->fork_ev = nullptr;
        if (m->own_stream && m->fork_ev && hipEventRecord(m->fork_ev, st) == hipSuccess && hipStreamWaitEvent(m->own_stream, m->fork_ev, 0) == hipSuccess)
            st = m->own_stream;
    }
#endif
    const mg_ocr_config& c = m->c;
    const int cap = round_up(L + max_new_tokens, 64), T_cap = round_up(L, 64);
    mg_memset_async(w.counters, 0, 16 * sizeof(int), st);
    ocr_init(out_ids, w.unfinished, w.counters, B, max_new_tokens, c.pad_token_id, st);
    if ((patch_pos == nullptr) != (patch_mask == nullptr)) return failf(MG_E_ARG, "mg_ocr_generate: patch_pos and patch_mask go together");
    if (pixel_values && n_img > 0) image_features(m, w, pixel_values, patch_pos, patch_mask, B * n_img, st);
    prefill(m, w, input_ids, (pixel_values && n_img > 0) ? w.feats : nullptr, B, n_img, L, cap, st);
    // logits of the last prompt position
    rmsnorm_pack_rows(w.h, m->rawp("model.text_model.norm.weight"), w.xc, w.last_rows, B * T_cap, c.t_hidden, c.rms_eps, 1.0f, st);
    GemmArgs lg = ga(w.xc, m->at<uint16_t>(m->lm_head), B, c.vocab, c.t_hidden);
    lg.out_f32 = w.logits; lg.ldo = c.vocab;
    gemm_rows(lg, EPI_F32_STORE, st);
    int host_flag[4] = {0, 0, 0, 0};
    int steps = 0;
    auto select = [&](int t, const int* pos_dev) {
        ArgmaxArgs g{};
        g.logits = w.logits; g.rows = B; g.V = c.vocab; g.ldl = c.vocab; g.eos = c.eos_token_id; g.pad = c.pad_token_id;
        g.next_ids = w.next_ids; g.out_ids = out_ids; g.max_len = max_new_tokens; g.pos = pos_dev ? 0 : t; g.pos_dev = pos_dev; g.min_len = 0;
        g.unfinished = w.unfinished; g.n_unfinished = w.counters + 5; g.step_ctr = w.counters;
        g.n_eos_more = c.n_eos_extra;
        for (int k = 0; k < c.n_eos_extra; ++k) g.eos_more[k] = c.eos_extra[k];
        greedy_select(g, st);
    };
    bool graphed = false;
#ifndef MG_EMU
    if (m->use_graph == 1 && !step_logits && max_new_tokens > 1) {
        const mg_ocr_model::Key key{ws, out_ids, (const void*)st, B, n_img, L, max_new_tokens};
        if (!(m->gvalid && m->gkey == key)) {
            std::lock_guard<std::mutex> capture_lock(mg_capture_mutex());
            m->greset();
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                decode_step(m, w, B, L - 1, w.counters + 2, cap, st);   // position = L - 1 + step counter (>= 1 here)
                select(0, w.counters + 2);
                if (hipStreamEndCapture(st, &graph) == hipSuccess && graph &&
                    hipGraphInstantiate(&m->gexec, graph, nullptr, nullptr, 0) == hipSuccess) { m->gkey = key; m->gvalid = true; }
                if (graph) (void)hipGraphDestroy(graph);
            }
            (void)hipGetLastError();
        }
        graphed = m->gvalid;
    }
#endif
    m->graph_active = graphed;
    if (step_logits && capture_steps > 0)
        MG_LAUNCH(copy_f32_kernel, dim3(256), dim3(256), 0, st, (const float*)w.logits, step_logits, (size_t)B * c.vocab);
    select(0, nullptr);
    steps = 1;
    for (int t = 1; t < max_new_tokens; ++t) {
        if ((t & 7) == 0) {     // termination is looked at every 8 steps: overrunning only appends pad columns (trimmed below)
            mg_memcpy_async(host_flag, w.counters, sizeof host_flag, st);
            mg_stream_sync(st);
            if (host_flag[0] == 0) break;
        }
#ifndef MG_EMU
        if (graphed) {
            if (hipGraphLaunch(m->gexec, st) != hipSuccess) return failf(MG_E_HIP, "mg_ocr_generate: hipGraphLaunch failed");
        } else
#endif
        {
            decode_step(m, w, B, L + t - 1, nullptr, cap, st);
            if (step_logits && t < capture_steps)
                MG_LAUNCH(copy_f32_kernel, dim3(256), dim3(256), 0, st, (const float*)w.logits, step_logits + (size_t)t * B * c.vocab, (size_t)B * c.vocab);
            select(t, nullptr);
        }
        steps = t + 1;
    }
    mg_memcpy_async(host_flag, w.counters, sizeof host_flag, st);
    mg_stream_sync(st);
    rc = check("mg_ocr_generate");
    if (rc != MG_OK) return rc;
    if (host_flag[3]) return failf(MG_E_INPUT, "mg_ocr_generate: %d bad inputs (token id outside the vocabulary, or a sequence whose <image> count differs from n_img * %d)", host_flag[3], m->T_img);
    *out_cols_host = host_flag[1] >= 0 ? host_flag[1] + 1 : steps;
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Queue form of generate (continuous decoding, as mg_generate_stream for the main model): N pages, `slots` decode rows.  The
// vision tower + prompt prefill run first, `chunk` pages at a time, into per-PAGE KV caches [layer][N][kv heads][cap][64] (29 MB
// per page at the SmolDocling geometry with 1 200 positions: 512 pages = 15 GB of the 288 GB), each prefill also selecting its
// page's first token; then `slots` rows work through the pages: a row whose page emits a stop token (or max_new_tokens) takes the
// next page - a pointer change in the slot table, its cache is already there.  A call of the batch form walks every row to the
// longest page's length (OCR outputs of 10-120 cells differ by 10x); here the step count is sum(lengths) / slots.  Greedy ids per
// page equal the batch form's (rows are independent).  out_ids [N][max_new_tokens] (pad after the stop token), out_len [N] device.
int mg_ocr_stream_workspace_bytes(const mg_ocr_model* m, int N, int n_img, int L, int max_new_tokens, int slots, int chunk, size_t* out_bytes) {
    if (!m || !out_bytes || N < 1 || slots < 1 || chunk < 1) return failf(MG_E_ARG, "mg_ocr_stream_workspace_bytes: bad argument");
    Ws a, b;
    carve(m, nullptr, chunk, n_img, L, 0, false, &a);
    carve(m, nullptr, slots, 0, 1, 0, false, &b);
    const int cap = round_up(L + max_new_tokens, 64);
    const size_t kv = (size_t)N * m->c.t_kv_heads * cap * 64 * m->c.t_layers * sizeof(uint16_t);
    *out_bytes = align_up(a.total, 256) + align_up(b.total, 256) + 2 * align_up(kv, 256) + align_up((size_t)N * 8, 256) + 8 * 4096 + align_up((size_t)N * 4, 256);
    return MG_OK;
}

int mg_ocr_generate_stream(mg_ocr_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* pixel_values,
                           const int32_t* patch_pos, const uint8_t* patch_mask, int N, int n_img, int L, int max_new_tokens, int slots, int chunk,
                           int64_t* out_ids, int32_t* out_len, long* steps_host) {
    return mg_ocr_generate_stream_ragged(m, stream, ws, ws_bytes, input_ids, nullptr, pixel_values, patch_pos, patch_mask, N, n_img, L, max_new_tokens, slots, chunk,
                                         out_ids, out_len, steps_host);
}

int mg_ocr_generate_stream_ragged(mg_ocr_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const int32_t* prompt_len,
                                  const float* pixel_values, const int32_t* patch_pos, const uint8_t* patch_mask, int N, int n_img, int L, int max_new_tokens,
                                  int slots, int chunk, int64_t* out_ids, int32_t* out_len, long* steps_host) {
    int rc = check_args(m, chunk < N ? chunk : N, n_img, L, "mg_ocr_generate_stream");
    if (rc != MG_OK) return rc;
    std::unique_lock<std::recursive_mutex> call_lock;
    if (m) {
        call_lock = std::unique_lock<std::recursive_mutex>(m->call_mu, std::try_to_lock);
        if (!call_lock.owns_lock()) return failf(MG_E_STATE, "mg_ocr_generate_stream: this execution context is inside another call (one call at a time per context; mg_ocr_clone gives further contexts)");
    }
    if (N < 1 || slots < 1 || slots > 256 || chunk < 1 || chunk > 256) return failf(MG_E_SHAPE, "mg_ocr_generate_stream: N >= 1, slots and chunk in [1, 256]");
    if (max_new_tokens < 1 || !out_ids || !out_len) return failf(MG_E_ARG, "mg_ocr_generate_stream: bad output arguments");
    if (L + max_new_tokens > mg_ocr_model::MAX_POS) return failf(MG_E_SHAPE, "mg_ocr_generate_stream: %d + %d positions exceed %d", L, max_new_tokens, mg_ocr_model::MAX_POS);
    if ((patch_pos == nullptr) != (patch_mask == nullptr)) return failf(MG_E_ARG, "mg_ocr_generate_stream: patch_pos and patch_mask go together");
    size_t need = 0;
    mg_ocr_stream_workspace_bytes(m, N, n_img, L, max_new_tokens, slots, chunk, &need);
    if (!ws || ws_bytes < need) return failf(MG_E_WORKSPACE, "mg_ocr_generate_stream: workspace %zu < %zu bytes", ws_bytes, need);
    mgStream_t st = (mgStream_t)stream;
#ifndef MG_EMU
    if (m->use_graph == 1 && st == nullptr) {
        if (!m->own_stream && hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess) m->own_stream = nullptr;
        if (!m->fork_ev && hipEventCreateWithFlags(&m->fork_ev, hipEventDisableTiming) != hipSuccess) m->fork_ev = nullptr;
        if (m->own_stream && m->fork_ev && hipEventRecord(m->fork_ev, st) == hipSuccess && hipStreamWaitEvent(m->own_stream, m->fork_ev, 0) == hipSuccess)
            st = m->own_stream;
    }
#endif
    const mg_ocr_config& c = m->c;
    const int cap = round_up(L + max_new_tokens, 64), T_cap = round_up(L, 64);
    // workspace: [prefill chunk | decode rows | K pages | V pages | first tokens | slot table]
    char* base = (char*)ws;
    Ws wp, wd;
    carve(m, base, chunk, n_img, L, 0, false, &wp);
    base += align_up(wp.total, 256);
    carve(m, base, slots, 0, 1, 0, false, &wd);
    base += align_up(wd.total, 256);
    const size_t page_kv = (size_t)c.t_kv_heads * cap * 64;              // elements of one page's keys (or values) in one layer
    const size_t kv_layer = (size_t)N * page_kv;
    uint16_t* Kbig = (uint16_t*)base; base += align_up(kv_layer * c.t_layers * sizeof(uint16_t), 256);
    uint16_t* Vbig = (uint16_t*)base; base += align_up(kv_layer * c.t_layers * sizeof(uint16_t), 256);
    int64_t* first_tok = (int64_t*)base; base += align_up((size_t)N * 8, 256);
    int* spos = (int*)base; base += 4096;
    int* simg = (int*)base; base += 4096;
    int* spool = (int*)base; base += 4096;
    int* sctr = (int*)base; base += 4096;
    int* scratch_unf = (int*)base; base += 4096;          // prefill-side selection: flags / counter it needs but nobody reads
    int* err = (int*)base; base += 4096;
    int* toff = (int*)base; base += align_up((size_t)N * 4, 256);          // per page: prompt length - L (prompts of different lengths)
    mg_memset_async(sctr, 0, 4096, st);
    mg_memset_async(err, 0, 4096, st);
    if (prompt_len) ocr_len_delta(prompt_len, toff, N, L, err, st);
    mg_memset_async(out_len, 0, (size_t)N * sizeof(int32_t), st);
    ocr_init(out_ids, scratch_unf, sctr + 512, N < 1024 ? N : 1024, max_new_tokens, c.pad_token_id, st);      // (pads the first rows; the rest below)
    if (N > 1024) for (int r0 = 1024; r0 < N; r0 += 1024)
        ocr_init(out_ids + (size_t)r0 * max_new_tokens, scratch_unf, sctr + 512, (N - r0) < 1024 ? (N - r0) : 1024, max_new_tokens, c.pad_token_id, st);
    ocr_slots_init(wd.unfinished, spos, simg, spool, wd.next_ids, slots, sctr, N, st);
    const size_t img_in = (size_t)n_img * 3 * c.image_size * c.image_size;
    const int P = m->P;
    // 1. vision tower + prefill of every page (chunks), first token of every page
    for (int c0 = 0; c0 < N; c0 += chunk) {
        const int n = (N - c0) < chunk ? (N - c0) : chunk;
        Ws w;
        carve(m, (char*)ws, n, n_img, L, 0, false, &w);            // the chunk's own carving (a short last chunk uses less of the region)
        mg_memset_async(w.counters, 0, 16 * sizeof(int), st);
        if (pixel_values && n_img > 0)
            image_features(m, w, pixel_values + (size_t)c0 * img_in, patch_pos ? patch_pos + (size_t)c0 * n_img * P : nullptr,
                           patch_mask ? patch_mask + (size_t)c0 * n_img * P : nullptr, n * n_img, st);
        Ws wc = w;
        wc.Kc = Kbig + (size_t)c0 * page_kv; wc.Vc = Vbig + (size_t)c0 * page_kv; wc.kv_layer = kv_layer;
        prefill(m, wc, input_ids + (size_t)c0 * L, (pixel_values && n_img > 0) ? w.feats : nullptr, n, n_img, L, cap, st, prompt_len ? prompt_len + c0 : nullptr);
        rmsnorm_pack_rows(w.h, m->rawp("model.text_model.norm.weight"), w.xc, w.last_rows, n * T_cap, c.t_hidden, c.rms_eps, 1.0f, st);
        GemmArgs lg = ga(w.xc, m->at<uint16_t>(m->lm_head), n, c.vocab, c.t_hidden);
        lg.out_f32 = w.logits; lg.ldo = c.vocab;
        gemm_rows(lg, EPI_F32_STORE, st);
        ArgmaxArgs g{};          // batch-mode selection of column 0: token -> first_tok[page] and out_ids[page][0]
        g.logits = w.logits; g.rows = n; g.V = c.vocab; g.ldl = c.vocab; g.eos = c.eos_token_id; g.pad = c.pad_token_id;
        g.next_ids = first_tok + c0; g.out_ids = out_ids + (size_t)c0 * max_new_tokens; g.max_len = max_new_tokens; g.pos = 0; g.min_len = 0;
        ocr_fill_ints(scratch_unf, 1, n, st);
        g.unfinished = scratch_unf; g.n_unfinished = sctr + 600;
        g.n_eos_more = c.n_eos_extra;
        for (int k = 0; k < c.n_eos_extra; ++k) g.eos_more[k] = c.eos_extra[k];
        greedy_select(g, st);
        ocr_add_int(err, w.counters + 3, st);
    }
    ocr_set_int(sctr + 5, N, st);              // every page's cache is ready: the slots may take them all
    // 2. the slots
    SlotTable tab{};
    tab.pos = spos; tab.img = simg; tab.pool = spool; tab.ctr = sctr; tab.out_len = out_len; tab.pool_cap = N; tab.start_id = 0;
    tab.first_tok = first_tok; tab.n_stop = 1 + c.n_eos_extra; tab.stop[0] = c.eos_token_id;
    for (int k = 0; k < c.n_eos_extra; ++k) tab.stop[1 + k] = c.eos_extra[k];
    tab.max_len = max_new_tokens;
    Ws w = wd;
    w.Kc = Kbig; w.Vc = Vbig; w.kv_layer = kv_layer;
    const SlotView view{spos, spool, wd.unfinished, prompt_len ? toff : nullptr};
    auto step = [&]() {
        decode_step(m, w, slots, L, nullptr, cap, st, &view);
        ArgmaxArgs g{};
        g.logits = w.logits; g.rows = slots; g.V = c.vocab; g.ldl = c.vocab; g.eos = c.eos_token_id; g.pad = c.pad_token_id;
        g.next_ids = w.next_ids; g.out_ids = out_ids; g.max_len = max_new_tokens; g.min_len = 0;
        g.unfinished = w.unfinished; g.n_unfinished = sctr + 600;
        g.n_eos_more = c.n_eos_extra;
        for (int k = 0; k < c.n_eos_extra; ++k) g.eos_more[k] = c.eos_extra[k];
        g.slots = tab;
        greedy_select(g, st);
        slot_refill(tab, w.next_ids, w.unfinished, slots, st);
    };
    slot_refill(tab, w.next_ids, w.unfinished, slots, st);          // the first pages take their slots
    bool graphed = false;
#ifndef MG_EMU
    if (m->use_graph == 1) {
        const mg_ocr_model::SKey key{ws, out_ids, out_len, (const void*)st, N, slots, L, max_new_tokens, chunk, n_img, prompt_len ? 1 : 0};
        if (!(m->svalid && m->skey == key)) {
            std::lock_guard<std::mutex> capture_lock(mg_capture_mutex());
            m->sreset();
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                step();
                if (hipStreamEndCapture(st, &graph) == hipSuccess && graph && hipGraphInstantiate(&m->sexec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    m->skey = key; m->svalid = true;
                }
                if (graph) (void)hipGraphDestroy(graph);
            }
            (void)hipGetLastError();
        }
        graphed = m->svalid;
    }
#endif
    m->graph_active = graphed;
    long steps = 0;
    int host[16] = {0};
    const long limit = (long)N * max_new_tokens / 1 + 64;
    while (host[1] < N) {
        for (int g8 = 0; g8 < 8; ++g8) {
#ifndef MG_EMU
            if (graphed) { if (hipGraphLaunch(m->sexec, st) != hipSuccess) return failf(MG_E_HIP, "mg_ocr_generate_stream: hipGraphLaunch failed"); }
            else
#endif
                step();
            ++steps;
        }
        mg_memcpy_async(host, sctr, sizeof host, st);
        mg_stream_sync(st);
        if (steps > limit) return failf(MG_E_HIP, "mg_ocr_generate_stream: no progress (%d of %d pages after %ld steps)", host[1], N, steps);
    }
    int bad = 0;
    mg_memcpy_async(&bad, err, sizeof(int), st);
    mg_stream_sync(st);
    rc = check("mg_ocr_generate_stream");
    if (rc != MG_OK) return rc;
    if (steps_host) *steps_host = steps;
    if (bad) return failf(MG_E_INPUT, "mg_ocr_generate_stream: %d bad inputs (token id outside the vocabulary, a sequence whose <image> count differs from n_img * %d, or a prompt length outside [1, L])", bad, m->T_img);
    return MG_OK;
}

}  // extern "C"
