// OCSR vision branch "e1" (SURVEY.md §8 rows a7 / f-2): the C ABI `mg_e1_*` of include/mgrapher.h.  What the reference's model holds as
//   model.encoder.molscribe_encoder    MolScribe's Swin-B (timm swin_base_patch4_window12_384), loaded by model.init_molscribe_weights()
//   model.encoder.molscribe_projector  an MLP projector into d_model
// (ref: markushgrapher/core/common/begin.py:137-151, utils/model/utils_model_loading.py:20-36, README.md:212-215) and evaluates inside
// forward() / generate() of its transformers fork: pixel_values -> Swin features [B, 144, 1024] -> projector -> e1 [B, 144, d_model],
// concatenated with the VTL encoder's states in front of the decoder.  The Swin arithmetic follows stock transformers
// models/swin/modeling_swin.py (pinned: tests/golden/swin_*.npz); input derivation and projector are INFERRED (e1_shapes.py).
// Host code only orchestrates launches: contractions on the main path's GEMM kernels, glue and window attention in k_swin.hip.
// mg_e1_encode reads the model and writes only caller buffers: one mg_e1_model may serve several execution contexts at once.
#include "mg_kernels.h"
#include "mg_swin.h"
#include "../../include/mgrapher.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

using namespace mg;

namespace mg {
int e1_encode_nested(const mg_e1_model* m, mgStream_t st, void* ws, size_t ws_bytes, const float* pixel_values, int B, float* e1_out, float* features_out);
void ocr_pack_aug(const float* W, const float* bias, float scale, uint16_t* dst, int row0, int N, int K, int Kaug, int Nfill, int rstride, mgStream_t st);
}

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
struct Block { std::string p; size_t wqkv, bqkv, wo, w1, w2, tab; };
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

struct mg_e1_model {
    mg_e1_config c;
    int ns = 0, g = 0, Kp = 0, M_out = 0, C_out = 0, d_model = 0;
    int dim[4] = {0, 0, 0, 0}, res[4] = {0, 0, 0, 0};
    char* arena = nullptr;
    size_t arena_bytes = 0;
    std::map<std::string, Raw> raw;          // canonical key (e1_shapes.py) -> fp32 copy in the arena
    std::vector<std::vector<Block>> blk;
    std::vector<size_t> wred;                // patch-merging reductions
    std::vector<size_t> pw;                  // projector weights
    size_t patch_w = 0;
    bool finalized = false;
    template <typename T> T* at(size_t off) const { return (T*)(arena + off); }
    const float* rawp(const std::string& k) const { return (const float*)(arena + raw.at(k).off); }
};

namespace {

struct Ws {
    float *pix, *ha, *rm, *feats;
    uint16_t *xim, *x, *qkv, *ctx, *y, *xf, *pa, *pb;
    size_t total;
};

void carve(const mg_e1_model* m, char* base, int B, Ws* w) {
    const mg_e1_config& c = m->c;
    Carver cv{base};
    const size_t I = c.image_size, M0 = (size_t)B * m->g * m->g, C0 = c.embed_dim;
    const int M0p = round_up((int)M0, 32);
    w->pix = cv.take<float>((size_t)B * c.num_channels * I * I);
    w->xim = cv.take<uint16_t>(pk_elems(M0p, m->Kp));
    w->ha = cv.take<float>((size_t)M0p * C0);          // the tiled residual stream (rows padded to whole 32-row tiles)
    w->rm = cv.take<float>(M0 * C0);                  // row-major outputs of the plain fp32-store GEMMs (patch embedding, merge reductions)
    // packed activations: the widest of every stage (rows shrink 4x, widths grow 2x per stage: stage 0 is the largest; rows padded to 32)
    size_t x_e = 0, q_e = 0, y_e = 0;
    for (int i = 0; i < m->ns; ++i) {
        const int Mi = round_up(B * m->res[i] * m->res[i], 32), Ci = m->dim[i];
        x_e = std::max(x_e, pk_elems(Mi, Ci));
        q_e = std::max(q_e, pk_elems(Mi, 3 * Ci));
        y_e = std::max(y_e, pk_elems(Mi, c.mlp_ratio * Ci));
        if (i + 1 < m->ns) x_e = std::max(x_e, pk_elems(round_up(Mi / 4, 32), 4 * Ci));
    }
    w->x = cv.take<uint16_t>(x_e);
    w->qkv = cv.take<uint16_t>(q_e);
    w->ctx = cv.take<uint16_t>(x_e);
    w->y = cv.take<uint16_t>(y_e);
    const int Mo = round_up(B * m->M_out, 32);
    w->feats = cv.take<float>((size_t)B * m->M_out * m->C_out);
    w->xf = cv.take<uint16_t>(pk_elems(Mo, m->C_out));
    int pmax = 64;
    for (int j = 0; j + 1 < c.n_proj; ++j) pmax = std::max(pmax, c.proj_dims[j]);
    w->pa = cv.take<uint16_t>(pk_elems(Mo, pmax));
    w->pb = cv.take<uint16_t>(pk_elems(Mo, pmax));
    w->total = align_up(cv.off, 256);
}

}  // namespace

extern "C" {

int mg_e1_create(const mg_e1_config* cfg, mg_e1_model** out) {
    if (!cfg || !out) return failf(MG_E_ARG, "mg_e1_create: null argument");
    const mg_e1_config& c = *cfg;
    if (c.n_stages < 1 || c.n_stages > 4) return failf(MG_E_SHAPE, "mg_e1_create: n_stages = %d (1 .. 4)", c.n_stages);
    if (c.num_channels < 1 || c.num_channels > 4 || c.patch_size < 1 || c.image_size % c.patch_size || c.mlp_ratio < 1)
        return failf(MG_E_SHAPE, "mg_e1_create: image_size %d / patch_size %d / num_channels %d / mlp_ratio %d", c.image_size, c.patch_size, c.num_channels, c.mlp_ratio);
    if (c.n_proj < 1 || c.n_proj > 4 || c.proj_act < 0 || c.proj_act > 1) return failf(MG_E_SHAPE, "mg_e1_create: projector of %d layers, activation %d", c.n_proj, c.proj_act);
    if (c.src_image_size < 1) return failf(MG_E_SHAPE, "mg_e1_create: src_image_size %d", c.src_image_size);
    mg_e1_model* m = new mg_e1_model();
    m->c = c; m->ns = c.n_stages; m->g = c.image_size / c.patch_size;
    m->Kp = round_up(c.num_channels * c.patch_size * c.patch_size, 64);
    for (int i = 0; i < m->ns; ++i) {
        m->dim[i] = c.embed_dim << i;
        m->res[i] = m->g >> i;
        const int C = m->dim[i], H = c.num_heads[i], R = m->res[i];
        // v1 limits, all reported: head dim 32 (every Swin checkpoint), widths the LayerNorm kernel is instantiated for, maps that are
        // whole windows and at least one window wide (the reference geometry: 96 / 48 / 24 / 12 with window 12; stock itself cannot run a
        // map smaller than its window with a bias table: modeling_swin.py:435-448), even maps in front of a merge
        if ((m->g >> i) << i != m->g || C != 32 * H || c.depths[i] < 1 || !swin_ln_supported(C) || !swin_ln_supported(4 * C) || R < c.window_size ||
            !swin_attention_supported(c.window_size, R, C, H) || (i + 1 < m->ns && (R & 1))) {
            delete m;
            return failf(MG_E_SHAPE, "mg_e1_create: stage %d (width %d, %d heads, map %d x %d, window %d) is outside the supported geometry (head dim 32, "
                         "width 64 .. 1024 a power of two, window 4 / 8 / 12, map a multiple of the window)", i, C, H, R, R, c.window_size);
        }
    }
    m->M_out = m->res[m->ns - 1] * m->res[m->ns - 1];
    m->C_out = m->dim[m->ns - 1];
    m->d_model = c.proj_dims[c.n_proj - 1];
    for (int j = 0; j < c.n_proj; ++j)
        if (c.proj_dims[j] < 32 || c.proj_dims[j] % 64) { delete m; return failf(MG_E_SHAPE, "mg_e1_create: projector width %d (multiples of 64)", c.proj_dims[j]); }
    size_t off = 0;
    auto add = [&](const std::string& k, std::vector<int64_t> shape) {
        size_t n = 1;
        for (int64_t s : shape) n *= (size_t)s;
        off = align_up(off, 256);
        m->raw[k] = Raw{off, shape, n};
        off += n * sizeof(float);
    };
    const int64_t tw2 = (int64_t)(2 * c.window_size - 1) * (2 * c.window_size - 1);
    add("swin.embeddings.patch_embeddings.projection.weight", {c.embed_dim, c.num_channels, c.patch_size, c.patch_size});
    add("swin.embeddings.patch_embeddings.projection.bias", {c.embed_dim});
    add("swin.embeddings.norm.weight", {c.embed_dim}); add("swin.embeddings.norm.bias", {c.embed_dim});
    for (int i = 0; i < m->ns; ++i) {
        const int64_t C = m->dim[i], H = c.num_heads[i], F = (int64_t)c.mlp_ratio * C;
        for (int j = 0; j < c.depths[i]; ++j) {
            const std::string p = "swin.encoder.layers." + std::to_string(i) + ".blocks." + std::to_string(j) + ".";
            for (const char* n : {"q_proj", "k_proj", "v_proj", "o_proj"}) {
                add(p + "attention." + n + ".weight", {C, C});
                add(p + "attention." + n + ".bias", {C});
            }
            add(p + "attention.relative_position_bias.relative_position_bias_table", {tw2, H});
            add(p + "layernorm_before.weight", {C}); add(p + "layernorm_before.bias", {C});
            add(p + "layernorm_after.weight", {C}); add(p + "layernorm_after.bias", {C});
            add(p + "mlp.fc1.weight", {F, C}); add(p + "mlp.fc1.bias", {F});
            add(p + "mlp.fc2.weight", {C, F}); add(p + "mlp.fc2.bias", {C});
        }
        if (i + 1 < m->ns) {
            const std::string p = "swin.encoder.layers." + std::to_string(i) + ".downsample.";
            add(p + "reduction.weight", {2 * C, 4 * C});
            add(p + "norm.weight", {4 * C}); add(p + "norm.bias", {4 * C});
        }
    }
    add("swin.layernorm.weight", {m->C_out}); add("swin.layernorm.bias", {m->C_out});
    {
        int64_t in = m->C_out;
        for (int j = 0; j < c.n_proj; ++j) {
            add("proj." + std::to_string(j) + ".weight", {c.proj_dims[j], in});
            add("proj." + std::to_string(j) + ".bias", {c.proj_dims[j]});
            in = c.proj_dims[j];
        }
    }
    // packed bf16 operands and derived tables built by mg_e1_finalize
    auto pk = [&](int N, int K) { off = align_up(off, 256); const size_t o = off; off += pk_elems(round_up(N, 32), K) * 2; return o; };
    auto f32 = [&](size_t n) { off = align_up(off, 256); const size_t o = off; off += n * sizeof(float); return o; };
    m->patch_w = pk(c.embed_dim, m->Kp);
    m->blk.resize(m->ns);
    for (int i = 0; i < m->ns; ++i) {
        const int C = m->dim[i], F = c.mlp_ratio * C;
        for (int j = 0; j < c.depths[i]; ++j) {
            Block b;
            b.p = "swin.encoder.layers." + std::to_string(i) + ".blocks." + std::to_string(j) + ".";
            b.wqkv = pk(3 * C, C); b.bqkv = f32(3 * (size_t)C); b.wo = pk(C, C); b.w1 = pk(F, C); b.w2 = pk(C, F);
            b.tab = f32((size_t)tw2 * c.num_heads[i]);
            m->blk[i].push_back(b);
        }
        if (i + 1 < m->ns) m->wred.push_back(pk(2 * C, 4 * C));
    }
    {
        int in = m->C_out;
        for (int j = 0; j < c.n_proj; ++j) { m->pw.push_back(pk(c.proj_dims[j], in)); in = c.proj_dims[j]; }
    }
    m->arena_bytes = align_up(off, 256);
    *out = m;
    return MG_OK;
}

void mg_e1_destroy(mg_e1_model* m) { delete m; }
size_t mg_e1_weights_bytes(const mg_e1_model* m) { return m ? m->arena_bytes : 0; }
int mg_e1_out_tokens(const mg_e1_model* m) { return m ? m->M_out : 0; }
int mg_e1_bind_weights(mg_e1_model* m, void* arena) {
    if (!m || !arena) return failf(MG_E_ARG, "mg_e1_bind_weights: null argument");
    m->arena = (char*)arena;
    m->finalized = false;
    return MG_OK;
}

int mg_e1_load_tensor(mg_e1_model* m, void* stream, const char* key, const void* src, int src_is_bf16, const int64_t* shape, int ndim) {
    if (!m || !key || !src) return failf(MG_E_ARG, "mg_e1_load_tensor: null argument");
    if (!m->arena) return failf(MG_E_STATE, "mg_e1_load_tensor: no weights arena bound");
    auto it = m->raw.find(key);
    if (it == m->raw.end()) return failf(MG_E_KEY, "mg_e1_load_tensor: unknown key '%s'", key);
    Raw& r = it->second;
    if ((size_t)ndim != r.shape.size()) return failf(MG_E_SHAPE, "mg_e1_load_tensor: '%s' has %d dims, expected %zu", key, ndim, r.shape.size());
    for (int i = 0; i < ndim; ++i)
        if (shape[i] != r.shape[i]) return failf(MG_E_SHAPE, "mg_e1_load_tensor: '%s' dim %d is %lld, expected %lld", key, i, (long long)shape[i], (long long)r.shape[i]);
    (void)mg_peek_error();
    mg_err_site() = MgErrSite{0, nullptr};
    convert_to_f32(src, src_is_bf16, (float*)(m->arena + r.off), r.n, (mgStream_t)stream);
    r.loaded = true;
    m->finalized = false;
    return check(key);
}

int mg_e1_finalize(mg_e1_model* m, void* stream) {
    if (!m || !m->arena) return failf(MG_E_STATE, "mg_e1_finalize: no model / arena");
    for (auto& kv : m->raw)
        if (!kv.second.loaded) return failf(MG_E_STATE, "mg_e1_finalize: tensor '%s' was not loaded", kv.first.c_str());
    (void)mg_peek_error();
    mg_err_site() = MgErrSite{0, nullptr};
    mgStream_t st = (mgStream_t)stream;
    const mg_e1_config& c = m->c;
    auto pack = [&](const std::string& k, size_t dst, int row0, int N, int K, int Kp, int Nfill) {
        ocr_pack_aug(m->rawp(k), nullptr, 1.0f, m->at<uint16_t>(dst), row0, N, K, Kp, Nfill, 1, st);
    };
    const int kp = c.num_channels * c.patch_size * c.patch_size;
    pack("swin.embeddings.patch_embeddings.projection.weight", m->patch_w, 0, c.embed_dim, kp, m->Kp, round_up(c.embed_dim, 32));
    const int tw2 = (2 * c.window_size - 1) * (2 * c.window_size - 1);
    for (int i = 0; i < m->ns; ++i) {
        const int C = m->dim[i], F = c.mlp_ratio * C;
        for (const Block& b : m->blk[i]) {
            pack(b.p + "attention.q_proj.weight", b.wqkv, 0, C, C, C, C);
            pack(b.p + "attention.k_proj.weight", b.wqkv, C, C, C, C, C);
            pack(b.p + "attention.v_proj.weight", b.wqkv, 2 * C, C, C, C, round_up(3 * C, 32) - 2 * C);
            const char* names[3] = {"attention.q_proj.bias", "attention.k_proj.bias", "attention.v_proj.bias"};
            for (int q = 0; q < 3; ++q) mg_memcpy_async(m->at<float>(b.bqkv) + (size_t)q * C, m->rawp(b.p + names[q]), (size_t)C * sizeof(float), st);
            pack(b.p + "attention.o_proj.weight", b.wo, 0, C, C, C, round_up(C, 32));
            pack(b.p + "mlp.fc1.weight", b.w1, 0, F, C, C, round_up(F, 32));
            pack(b.p + "mlp.fc2.weight", b.w2, 0, C, F, F, round_up(C, 32));
            swin_transpose_f32(m->rawp(b.p + "attention.relative_position_bias.relative_position_bias_table"), m->at<float>(b.tab), tw2, c.num_heads[i], st);
        }
        if (i + 1 < m->ns) pack("swin.encoder.layers." + std::to_string(i) + ".downsample.reduction.weight", m->wred[i], 0, 2 * C, 4 * C, 4 * C, round_up(2 * C, 32));
    }
    {
        int in = m->C_out;
        for (int j = 0; j < c.n_proj; ++j) { pack("proj." + std::to_string(j) + ".weight", m->pw[j], 0, c.proj_dims[j], in, in, round_up(c.proj_dims[j], 32)); in = c.proj_dims[j]; }
    }
    const int rc = check("mg_e1_finalize");
    if (rc == MG_OK) m->finalized = true;
    return rc;
}

int mg_e1_workspace_bytes(const mg_e1_model* m, int B, size_t* out_bytes) {
    if (!m || !out_bytes || B < 1) return failf(MG_E_ARG, "mg_e1_workspace_bytes: bad argument");
    Ws w;
    carve(m, nullptr, B, &w);
    *out_bytes = w.total;
    return MG_OK;
}

int mg_e1_encode(const mg_e1_model* m, void* stream, void* ws, size_t ws_bytes, const float* pixel_values, int B, float* e1_out, float* features_out) {
    (void)mg_peek_error();               // whatever is pending in the runtime's per-thread last-error slot was not caused by this call
    mg_err_site() = MgErrSite{0, nullptr};
    return mg::e1_encode_nested(m, (mgStream_t)stream, ws, ws_bytes, pixel_values, B, e1_out, features_out);
}

}  // extern "C"

namespace mg {
void e1_info(const mg_e1_model* m, int* tokens, int* d_model, int* src_image_size, int* channels, int* finalized) {
    *tokens = m->M_out; *d_model = m->d_model; *src_image_size = m->c.src_image_size; *channels = m->c.num_channels; *finalized = m->finalized ? 1 : 0;
}
int e1_encode_nested(const mg_e1_model* m, mgStream_t st, void* ws, size_t ws_bytes, const float* pixel_values, int B, float* e1_out, float* features_out) {
    if (!m || !ws || !pixel_values || (!e1_out && !features_out)) return failf(MG_E_ARG, "mg_e1_encode: null argument");
    if (!m->finalized) return failf(MG_E_STATE, "mg_e1_encode: mg_e1_finalize has not run");
    if (B < 1 || B > 4096) return failf(MG_E_SHAPE, "mg_e1_encode: B = %d out of range", B);
    Ws w;
    carve(m, (char*)ws, B, &w);
    if (ws_bytes < w.total) return failf(MG_E_WORKSPACE, "mg_e1_encode: workspace %zu < %zu bytes", ws_bytes, w.total);
    const mg_e1_config& c = m->c;
    // the branch's own input: bilinear resize of the VTL model's pixel_values + per-channel affine (INFERRED: e1_shapes.py)
    const float* pix = pixel_values;
    bool ident = c.src_image_size == c.image_size;
    for (int ch = 0; ch < c.num_channels && ch < 3; ++ch) ident = ident && c.pix_scale[ch] == 1.0f && c.pix_shift[ch] == 0.0f;
    if (!ident) {
        SwinPixAffine af{};
        for (int ch = 0; ch < 4; ++ch) { af.scale[ch] = ch < 3 ? c.pix_scale[ch] : 1.0f; af.shift[ch] = ch < 3 ? c.pix_shift[ch] : 0.0f; }
        swin_resize(pixel_values, w.pix, B, c.num_channels, c.src_image_size, c.image_size, af, st);
        pix = w.pix;
    }
    // patch embedding (stock:277-286) + LayerNorm (stock:229).  The residual stream h lives in the TILED fp32 layout (ht_off) of the
    // main encoder: the residual projections (o_proj, fc2) update it through the batched 16-byte epilogue EPI_RESID_NORM (no gain, no
    // partial sums: LayerNorm needs the mean as well and stays a kernel of its own), the LayerNorm kernel reads 512-byte runs of it.
    // Plain fp32-store GEMMs (patch embedding, merge reductions) leave row-major rows in `rm`; the LayerNorm that follows converts.
    const int M0 = B * m->g * m->g, C0 = c.embed_dim;
    swin_im2col_pack(pix, w.xim, B, c.num_channels, c.image_size, c.patch_size, m->Kp, st);
    float* h = w.ha;
    {
        GemmArgs pe = ga(w.xim, m->at<uint16_t>(m->patch_w), M0, C0, m->Kp);
        pe.out_f32 = w.rm; pe.ldo = C0; pe.bias = m->rawp("swin.embeddings.patch_embeddings.projection.bias");
        gemm(pe, EPI_F32_STORE, st);
        SwinLnArgs n{};
        n.h_in = w.rm; n.in_tiled = 0; n.h_out = h; n.h_out_norm = 1;
        n.w = m->rawp("swin.embeddings.norm.weight"); n.b = m->rawp("swin.embeddings.norm.bias"); n.M = M0; n.C = C0;
        n.eps = 1e-5f;                                                  // nn.LayerNorm default (stock:183), not config.layer_norm_eps
        swin_layernorm(n, st);
    }
    bool from_rm = false;                          // the stage's rows still sit row-major in w.rm (after a merge)
    for (int i = 0; i < m->ns; ++i) {
        const int C = m->dim[i], H = c.num_heads[i], R = m->res[i], M = B * R * R, F = c.mlp_ratio * C;
        for (size_t j = 0; j < m->blk[i].size(); ++j) {
            const Block& b = m->blk[i][j];
            const int shift = ((j & 1) && R > c.window_size) ? c.window_size / 2 : 0;      // stock:650, 576-582
            SwinLnArgs n1{};                       // x = LN1(h);  h += o_proj.bias (the projection below adds its product)
            n1.h_in = from_rm ? w.rm : h; n1.in_tiled = from_rm ? 0 : 1; n1.h_out = h;
            n1.w = m->rawp(b.p + "layernorm_before.weight"); n1.b = m->rawp(b.p + "layernorm_before.bias");
            n1.add_bias = m->rawp(b.p + "attention.o_proj.bias"); n1.x_pk = w.x; n1.M = M; n1.C = C; n1.eps = c.layer_norm_eps;
            swin_layernorm(n1, st);
            from_rm = false;
            GemmArgs q = ga(w.x, m->at<uint16_t>(b.wqkv), M, 3 * C, C);
            q.out_pk = w.qkv; q.bias = m->at<float>(b.bqkv);
            gemm(q, EPI_PK_BIAS, st);
            SwinAttnArgs t{};
            t.qkv = w.qkv; t.ctx = w.ctx; t.table = m->at<float>(b.tab); t.B = B; t.R = R; t.C = C; t.H = H; t.w = c.window_size; t.shift = shift;
            swin_attention(t, st);
            GemmArgs o = ga(w.ctx, m->at<uint16_t>(b.wo), M, C, C);
            o.out_f32 = h;
            gemm(o, EPI_RESID_NORM, st);
            SwinLnArgs n2{};                       // x = LN2(h);  h += fc2.bias
            n2.h_in = h; n2.in_tiled = 1; n2.h_out = h; n2.w = m->rawp(b.p + "layernorm_after.weight"); n2.b = m->rawp(b.p + "layernorm_after.bias");
            n2.add_bias = m->rawp(b.p + "mlp.fc2.bias"); n2.x_pk = w.x; n2.M = M; n2.C = C; n2.eps = c.layer_norm_eps;
            swin_layernorm(n2, st);
            GemmArgs f1 = ga(w.x, m->at<uint16_t>(b.w1), M, F, C);
            f1.out_pk = w.y; f1.bias = m->rawp(b.p + "mlp.fc1.bias");
            gemm(f1, EPI_PK_GELU_ERF, st);
            GemmArgs f2 = ga(w.y, m->at<uint16_t>(b.w2), M, C, F);
            f2.out_f32 = h;
            gemm(f2, EPI_RESID_NORM, st);
        }
        if (i + 1 < m->ns) {                       // patch merging (stock:309-326): gather 2 x 2, LayerNorm(4C) (nn.LayerNorm default eps), Linear(4C -> 2C) without bias
            const std::string p = "swin.encoder.layers." + std::to_string(i) + ".downsample.";
            SwinLnArgs n{};
            n.h_in = h; n.in_tiled = 1; n.w = m->rawp(p + "norm.weight"); n.b = m->rawp(p + "norm.bias"); n.x_pk = w.x; n.M = M / 4; n.C = 4 * C; n.merge_R = R; n.eps = 1e-5f;
            swin_layernorm(n, st);
            GemmArgs r = ga(w.x, m->at<uint16_t>(m->wred[i]), M / 4, 2 * C, 4 * C);
            r.out_f32 = w.rm; r.ldo = 2 * C;
            gemm(r, EPI_F32_STORE, st);
            from_rm = true;
        }
    }
    // final LayerNorm (stock:885-887) = SwinModel.last_hidden_state
    const int Mo = B * m->M_out;
    {
        SwinLnArgs n{};
        n.h_in = from_rm ? w.rm : h; n.in_tiled = from_rm ? 0 : 1;
        n.w = m->rawp("swin.layernorm.weight"); n.b = m->rawp("swin.layernorm.bias"); n.x_pk = w.xf; n.out_f32 = features_out ? features_out : w.feats;
        n.M = Mo; n.C = m->C_out; n.eps = c.layer_norm_eps;
        swin_layernorm(n, st);
    }
    // projector (INFERRED: Linear / GELU stack)
    if (e1_out) {
        const uint16_t* x = w.xf;
        int in = m->C_out;
        for (int j = 0; j < c.n_proj; ++j) {
            const int N = c.proj_dims[j];
            GemmArgs a = ga(x, m->at<uint16_t>(m->pw[j]), Mo, N, in);
            a.bias = m->rawp("proj." + std::to_string(j) + ".bias");
            if (j + 1 == c.n_proj) {
                a.out_f32 = e1_out; a.ldo = N;
                gemm(a, EPI_F32_STORE, st);
            } else {
                uint16_t* dst = (j & 1) ? w.pb : w.pa;
                a.out_pk = dst;
                gemm(a, c.proj_act == 1 ? EPI_PK_GELU_ERF : EPI_PK_BIAS, st);
                x = dst;
            }
            in = N;
        }
    }
    return check("mg_e1_encode");
}
}  // namespace mg
