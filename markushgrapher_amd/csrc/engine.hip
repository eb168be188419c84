// Engine: the C ABI of include/mgrapher.h on top of the HIP kernels — weight arena layout and loading, workspace
// carving, the VTL encoder, the teacher-forced decoder, and the KV-cached greedy / beam-search generate loop.
// Host code only orchestrates launches on the caller's stream; every byte of model arithmetic runs in HIP kernels.
#include "mg_kernels.h"
#include "../../include/mgrapher.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace mg;

namespace {

thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace
namespace mg {
int fail_msg(int code, const char* msg) { g_err = msg; return code; }      // ocr.hip
// swin.hip: the OCSR vision branch evaluated inside another entry point (no error-state reset), and what mg_attach_e1 checks
int e1_encode_nested(const mg_e1_model* m, mgStream_t st, void* ws, size_t ws_bytes, const float* pixel_values, int B, float* e1_out, float* features_out);
void e1_info(const mg_e1_model* m, int* tokens, int* d_model, int* src_image_size, int* channels, int* finalized);
}
namespace {
size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
int round_up(int x, int a) { return (x + a - 1) / a * a; }
constexpr int MG_ABSORB_AUTO_ROWS = 96;
bool use_absorb(const mg_model* m, int K, int rows);

struct EncLayer { size_t wqkv, wo, ln0, wi, wo2, ln1; };
// xq2 / wi2: product weights of the decode step (built by mg_finalize): [Wxq·G1 | Wxq·G1·Wo] and [Wi·G2 | Wi·G2·Wxo]
// xwk / xwv: absorbed cross-attention weights (k_xattn.hip): Wk_h feature-major, Wv_h in fragment order
struct DecLayer { size_t wqkv, wo, ln0, xq, xkv, xo, ln1, wi, wo2, ln2, xq2, wi2, xwk, xwv; };

}  // namespace

// A captured decode step.  Every step-dependent value (cache position, key count, output column) is read from the
// device step counter, so one executable graph serves all steps of a call — and later calls with the same buffers.
struct StepGraph {
    struct Key {
        const void *ws, *out_ids, *top2, *stream;
        int B, L, K, max_length, min_length, early_stopping, M_e1;
        float length_penalty;
        const void* scores;      // beam queue: beam_slot_end_kernel holds the out_scores pointer (NULL or a buffer) inside the captured launch
        bool operator==(const Key& o) const {
            return ws == o.ws && out_ids == o.out_ids && top2 == o.top2 && stream == o.stream && B == o.B && L == o.L && K == o.K && M_e1 == o.M_e1 &&
                   max_length == o.max_length && min_length == o.min_length && early_stopping == o.early_stopping &&
                   length_penalty == o.length_penalty && scores == o.scores;
        }
    };
    Key key{};
    bool valid = false;
#ifndef MG_EMU
    hipGraphExec_t exec = nullptr;
    void reset() { if (exec) (void)hipGraphExecDestroy(exec); exec = nullptr; valid = false; }
#else
    void reset() { valid = false; }
#endif
};

struct mg_model {
    mg_config c;
    int d, H, inner, dff, V, P, n_side, Kpatch, M2, T_cap;
    char* arena = nullptr;
    size_t arena_bytes = 0;
    size_t tok_emb, lm_head, patch_w, patch_b, x_emb, y_emb, rb_raw[3], rb_dec_raw, dec_tab;
    size_t bk1, bkhv, bkdec, enc_ln, dec_ln;
    std::vector<EncLayer> enc;
    std::vector<DecLayer> dec;
    std::map<std::string, bool> loaded;
    bool lm_head_loaded = false, finalized = false;
    std::vector<int> h_bk1, h_bkhv, h_bkdec;
    // encoder state left in the workspace by the last mg_encode
    int st_B = 0, st_L = 0, st_S = 0, st_Scap = 0, st_M = 0;
    bool st_row_tiles = false;          // the last mg_encode on this context ran its GEMMs on the live row-tile list (w.row_tiles is valid)
    void* st_ws = nullptr;
    // optional live timing of the dominant decode kernel (cross-attention K/V stream), HIP events on the caller's stream
    int prof_every = 0;
    std::vector<mgEvent_t> prof_ev;
    size_t prof_used = 0;
    double prof_ms = 0.0;
    double prof_empty_ms = 0.0;   // summed duration of the empty event brackets recorded right after each timed launch
    long prof_n = 0;
    double prof_keys = 0.0;   // sum over timed launches of the number of (image, key) pairs streamed
    // one decode step captured as a HIP graph (greedy and beam); replayed while its key matches the call
    size_t fin_a = 0, fin_b = 0, fin_c = 0;
    StepGraph step_graph;
    std::vector<float> beam_div_host;
    int use_graph = 1;
    bool graph_active = false;
    // Greedy decoding with the weight-absorbed cross-attention (k_xattn.hip): a layer streams the encoder states once instead of its K and V.
    // absorb: 2 (default where the geometry is supported) = by the call's decode rows (>= 96: absorbed), 1: every greedy call, 0: the K / V form for
    // every call (mg_set_cross_absorb, MG_XATTN_ABSORB).  Beam search keeps the K / V form.
    int xa_nt = 1;           // the stream's copies carry the non-temporal hint (states read once per layer by one CU must not displace the weights in L2 / MALL: +5 % in flight; MG_XATTN_NT=0)
    int absorb = 2, xa_split = 1, xa_stages = 4;      // (xa_stages: 4 = two wave groups, 136 KB of LDS: equal in flight, faster alone; 3 = one group, 100 KB - a decode projection's workgroup fits beside it on the CU)
    int shared_gpu = 0;       // mg_set_shared_gpu: other contexts run beside this one (the cross-attention stream keeps one workgroup per CU resident)
    // optional phase timing of mg_generate (HIP events): [start, encoder + cross-K/V done, decode loop done]
    bool phase_on = false;
    mgEvent_t phase_ev[3] = {};
    long phase_n = 0;
    double phase_enc_ms = 0.0, phase_dec_ms = 0.0;
    // parity-test instrumentation of the greedy decode loop (mg_debug_decode_capture)
    float* dbg_logits = nullptr;
    int dbg_steps = 0;
    const int64_t* dbg_forced = nullptr;
    // continuous decoding (mg_generate_stream): the encoder runs ahead on its own stream
    StepGraph stream_graph;
    int* stream_host = nullptr;          // pinned read-back ring of the stream counters
    int enc_mode = 1;                    // 0: encoder on the caller's stream (no overlap); 1: own low-priority stream; 2: own stream restricted to enc_mask
    std::vector<uint32_t> enc_mask;
    bool enc_stream_ready = false;
    mgStream_t enc_stream = nullptr;
    std::vector<mgEvent_t> chunk_ev;     // per pool chunk: encoder + cross-K/V of the chunk done
    std::vector<mgEvent_t> rb_ev;        // read-back ring
    std::vector<mgEvent_t> pace_ev;      // mg_generate: pacing of the launching host thread
    int pace = 1;                        // MG_PACE=0: enqueue as fast as the runtime accepts (the launching thread then spins on a full queue)
    mgEvent_t start_ev = nullptr;
    long stream_steps = 0, stream_idle_steps = 0;      // statistics of the last mg_generate_stream call
    double stream_enc_ms = 0.0;
    // encoder GEMMs skip 32-row tiles without an attended position (padded text slots, slots of dropped patches); MG_ENC_ROW_TILES=0
    // computes every row (A/B; results of attended rows are bit-identical either way)
    bool row_tiles = true;
    // trailing text padding of a batch: false = stock batched semantics (the padded slots sit between text and patches and count in the
    // 1-D position bias), true = per-image semantics (every image as if alone and unpadded: the reference's batch size is 1)
    bool trim_padding = false;
    bool fused_tail = true;    // MG_DECODE_FUSED_TAIL=0: separate embedding / selection launches (A/B; identical results)
    // one call at a time per execution context (mg_clone gives further contexts); recursive: mg_generate runs mg_encode
    std::recursive_mutex call_mu;
    bool tied = true;          // tie_word_embeddings: lm_head = shared.weight and logits scaled by d_model^-0.5 (stock:1554-1557)
    // OCSR vision branch (mg_attach_e1, swin.hip): calls that pass no precomputed e1 evaluate it from pixel_values themselves
    const mg_e1_model* e1m = nullptr;
    int e1_M = 0;              // its tokens per image
#ifndef MG_EMU
    hipStream_t own_stream = nullptr;
    hipEvent_t fork_ev = nullptr;
#endif
    ~mg_model() {
        step_graph.reset();
        stream_graph.reset();
        if (stream_host) mg_host_free(stream_host);
        if (enc_stream_ready && enc_stream) mg_stream_destroy(enc_stream);
        for (mgEvent_t e : chunk_ev) mg_event_destroy(e);
        for (mgEvent_t e : rb_ev) mg_event_destroy(e);
        for (mgEvent_t e : pace_ev) mg_event_destroy(e);
        if (start_ev) mg_event_destroy(start_ev);
#ifndef MG_EMU
        if (own_stream) (void)hipStreamDestroy(own_stream);
        if (fork_ev) (void)hipEventDestroy(fork_ev);
#endif
        for (mgEvent_t e : phase_ev) if (e) mg_event_destroy(e);
    }

    template <typename T> T* at(size_t off) const { return (T*)(arena + off); }
};

namespace {

// stock:422-468 bucket function on integer distances.  The reference evaluates
// max_exact + trunc(log(n/max_exact)/log(max_distance/max_exact)*(nb-max_exact)) in fp32; integer n sit exactly
// on a bucket edge only when n/max_exact is an exact power (e.g. 16, 32, 64 for max_distance 128), where torch
// yields the clean integer — reproduced here in long double with a tiny upward nudge, and pinned against the
// torch-evaluated golden table in tests/golden/bucket_tables.npz.
int bucket_of(long rel, bool bidirectional, int num_buckets, int max_distance) {
    int ret = 0;
    long n;
    if (bidirectional) {
        num_buckets /= 2;
        if (rel > 0) ret += num_buckets;
        n = rel < 0 ? -rel : rel;
    } else {
        n = rel < 0 ? -rel : 0;
    }
    const int max_exact = num_buckets / 2;
    if (n < max_exact) return ret + (int)n;
    long double v = logl((long double)n / max_exact) / logl((long double)max_distance / max_exact) * (num_buckets - max_exact);
    int large = max_exact + (int)floorl(v + 1e-9L);
    if (large > num_buckets - 1) large = num_buckets - 1;
    return ret + large;
}

__global__ __launch_bounds__(256) void build_table_kernel(const float* raw, const int* bucket, float* out, int n, int H) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * H; i += gridDim.x * blockDim.x)
        out[i] = raw[bucket[i / H] * H + (i % H)];
}

__global__ __launch_bounds__(256) void fill_ids_kernel(int64_t* next_ids, int64_t* out_ids, int* unfinished, int* counters, int rows,
                                                  int max_len, int64_t start, int64_t pad) {
    const int r = blockIdx.x;
    for (int j = threadIdx.x; j < max_len; j += blockDim.x) out_ids[(size_t)r * max_len + j] = (j == 0) ? start : pad;
    if (threadIdx.x == 0) {
        next_ids[r] = start;
        unfinished[r] = 1;
        if (r == 0) { counters[0] = rows; counters[1] = -1; counters[2] = 0; counters[5] = 0; counters[6] = 0; }   // [3] keeps the encoder's input-error count
    }
}
// counters: [0] n_unfinished, [1] done_step (first step after which every row had finished), [2] step, [3] input errors,
// [4] beam output columns, [5] greedy: unfinished rows counted by this step's selection (published to [0] by the last
// selection workgroup), [6] greedy: arrival counter of the selection workgroups
__global__ void step_end_kernel(int* counters, int greedy) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (greedy) { counters[0] = counters[5]; counters[5] = 0; }
        if (counters[0] == 0 && counters[1] < 0) counters[1] = counters[2];
        counters[2] += 1;
    }
}
// parity-test instrumentation (mg_debug_decode_capture): copy the step's logits / overwrite the token fed to the next step
__global__ __launch_bounds__(256) void capture_logits_kernel(const float* logits, int ldl, float* out, int rows, int V) {
    const size_t n = (size_t)rows * V;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / V, c = i - r * V;
        out[i] = logits[r * ldl + c];
    }
}
__global__ __launch_bounds__(64) void force_ids_kernel(int64_t* next_ids, const int64_t* forced, int rows, int max_len, int pos) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) next_ids[r] = forced[(size_t)r * max_len + pos];
}
// teacher-forced decoder: pad [B][T] ids to [B][T_cap] rows and build the compact-row map of the valid positions
__global__ __launch_bounds__(256) void pad_dec_inputs_kernel(const int64_t* ids, const uint8_t* mask, int64_t* ids_pad, uint8_t* mask_pad,
                                                        int* dst_row, int B, int T, int T_cap) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * T_cap; i += gridDim.x * blockDim.x) {
        const int b = i / T_cap, t = i - b * T_cap;
        const bool in = t < T;
        ids_pad[i] = in ? ids[(size_t)b * T + t] : 0;
        mask_pad[i] = in ? (mask ? (mask[(size_t)b * T + t] != 0) : 1) : 0;
        dst_row[i] = in ? b * T + t : -1;
    }
}
// Output position s of image b (the documented [text L | patches P] order) -> row of the workspace.  With per-image padding
// semantics (text_len[b] = Lb < L) the workspace holds [text Lb | patches P | text padding L - Lb].
MG_DEV size_t enc_src_row(int s, int L, int P, const int* text_len, int b) {
    const int Lb = text_len ? text_len[b] : L;
    if (s < Lb) return (size_t)s;
    if (s < L) return (size_t)(Lb + P + (s - Lb));
    return (size_t)(Lb + (s - L));
}
__global__ __launch_bounds__(256) void copy_rows_kernel(const float* src, float* dst, int B, int S, int S_cap, int d, int L, const int* text_len) {
    const size_t n = (size_t)B * S * (d / 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / (d / 4), c = i - row * (d / 4);
        const size_t b = row / S, s = row - b * S;
        ((float4*)dst)[i] = ((const float4*)src)[(b * S_cap + enc_src_row((int)s, L, S - L, text_len, (int)b)) * (d / 4) + c];
    }
}
__global__ __launch_bounds__(256) void copy_bytes_rows_kernel(const uint8_t* src, uint8_t* dst, int B, int S, int S_cap, int L, const int* text_len) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * S; i += gridDim.x * blockDim.x) {
        const int b = i / S, s = i - b * S;
        dst[i] = src[(size_t)b * S_cap + enc_src_row(s, L, S - L, text_len, b)];
    }
}

// ---- workspace -------------------------------------------------------------------------------------------
struct Ws {
    // encoder
    uint16_t *xim, *x_pk, *q_pk, *k_pk, *vt_pk, *ctx_pk, *y_pk, *enc_pk, *bidx;
    float *patch_emb, *hidden, *enc_f32, *enc_part_a, *enc_part_b;
    void* meta;
    double *cx, *cy;
    uint8_t* mask;
    int *xrow, *xlen, *counters, *att_kst, *row_tiles, *text_len;
    uint8_t* att_qbv;
    // OCSR-branch tokens e1, packed for the cross-K/V projections
    uint16_t* e1_pk;
    int* e1_map;
    uint8_t* xmask;           // teacher-forced cross-attention key mask over [e1 | encoder] positions
    float* e1_f32;            // attached e1 branch: its output [B][M][d] and its workspace
    char* e1_ws;
    size_t e1_ws_bytes;
    // decode (generate)
    uint16_t *xk, *xv, *sk, *sv, *dq, *dx_pk, *dy_pk;
    // weight-absorbed cross-attention (greedy): the states the decoder attends [B][Sx_cap][d], q' [rows][H][d], context partials
    uint16_t *encx, *qx, *xpart;
    float *xml;
    uint16_t *xa, *xb;        // packed [rows][d + inner] operand windows of the pair projections: [bf16(h) | attention context]
    float *dh, *logits, *slabs, *rs_part, *rs_part1, *rs_part2;
    float4* ptop;             // fused greedy tail: per-workgroup top-2 partials of the lm_head launch [rows][V/32]
    float* stopv;             // [rows][4] logits of the stop tokens
    size_t slab_stride;
    int* tickets;             // arrival counters of the K-slab projections (zero between launches)
    int64_t* next_ids;
    int *unfinished, *anc, *beam_idx;
    float* beam_div;          // [T_cap + 1] length-penalty divisor per cur_len
    void* beam_state;
    // teacher-forced decoder
    int64_t* tf_ids;
    uint8_t* tf_mask;
    int* tf_rowmap;
    float* tf_hidden;
    uint16_t *tf_x, *tf_q, *tf_k, *tf_vt, *tf_ctx, *tf_y, *tf_xk, *tf_xvt, *tf_xc;
    size_t total;
};

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

void carve(const mg_model* m, char* base, int B, int L, int K, int max_len, int T, int Me1, Ws* w) {
    Carver c{base};
    const int d = m->d, inner = m->inner, H = m->H;
    const int S_cap = round_up(L + m->P, 64);
    const int M64 = Me1 > 0 ? round_up(Me1, 64) : 0, Sx_cap = S_cap + M64;      // cross-attention keys: [e1 | encoder]
    const size_t M = (size_t)B * S_cap;
    const size_t MP = (size_t)round_up(B * m->P, 32);
    w->xim = c.take<uint16_t>(MP * m->Kpatch);
    w->patch_emb = c.take<float>(MP * d);
    w->meta = c.take<char>(embed_meta_bytes(B, S_cap));
    w->hidden = c.take<float>(M * d);
    w->x_pk = c.take<uint16_t>(M * d);
    w->q_pk = c.take<uint16_t>(M * inner);
    w->k_pk = c.take<uint16_t>(M * inner);
    w->vt_pk = c.take<uint16_t>(M * inner);
    w->ctx_pk = c.take<uint16_t>(M * inner);
    w->y_pk = c.take<uint16_t>(M * m->dff);
    w->enc_pk = c.take<uint16_t>(M * d);
    w->enc_f32 = c.take<float>(M * d);
    w->bidx = c.take<uint16_t>(M * (size_t)S_cap);
    {   // deferred-RMSNorm partial sums of the encoder: [rows][d/64 rounded up to 4] after the attention / FFN output
        const size_t np4 = (size_t)round_up(d / 64 > 0 ? d / 64 : 1, 4);
        w->enc_part_a = c.take<float>(M * np4);
        w->enc_part_b = c.take<float>(M * np4);
    }
    w->cx = c.take<double>(M);
    w->cy = c.take<double>(M);
    w->mask = c.take<uint8_t>(M);
    w->xrow = c.take<int>(M);
    w->xlen = c.take<int>(B);
    w->counters = c.take<int>(16);
    w->att_kst = c.take<int>((size_t)B * (1 + (S_cap >> 6)));
    w->att_qbv = c.take<uint8_t>((size_t)B * ((S_cap + 127) / 128));
    w->text_len = c.take<int>(B);
    w->row_tiles = c.take<int>(M / 32 + 1);           // [0] = count, then the live 32-row tiles (GemmArgs::row_tiles)
    w->e1_pk = c.take<uint16_t>((size_t)B * M64 * d);
    w->e1_map = c.take<int>((size_t)B * M64);
    w->xmask = c.take<uint8_t>((size_t)B * Sx_cap);
    w->e1_f32 = nullptr; w->e1_ws = nullptr; w->e1_ws_bytes = 0;
    if (m->e1m && Me1 > 0) {
        w->e1_f32 = c.take<float>((size_t)B * Me1 * d);
        (void)mg_e1_workspace_bytes(m->e1m, B, &w->e1_ws_bytes);
        w->e1_ws = c.take<char>(w->e1_ws_bytes);
    }
    if (max_len > 0) {
        const int R = B * K, Rp = round_up(R, 32);
        const size_t nl = m->dec.size();
        w->xk = w->xv = w->encx = w->qx = w->xpart = nullptr; w->xml = nullptr;
        if (use_absorb(m, K, R)) {
            w->encx = c.take<uint16_t>((size_t)B * Sx_cap * d);
            w->qx = c.take<uint16_t>((size_t)Rp * H * d);
            w->xpart = c.take<uint16_t>((size_t)Rp * m->xa_split * H * d);
            w->xml = c.take<float>((size_t)Rp * m->xa_split * H * 2);
        } else {
            w->xk = c.take<uint16_t>(nl * B * H * Sx_cap * 64);
            w->xv = c.take<uint16_t>(nl * B * H * Sx_cap * 64);
        }
        w->sk = c.take<uint16_t>(nl * R * H * (size_t)m->T_cap * 64);
        w->sv = c.take<uint16_t>(nl * R * H * (size_t)m->T_cap * 64);
        w->dq = c.take<uint16_t>((size_t)Rp * inner);
        w->dx_pk = c.take<uint16_t>((size_t)Rp * d);
        w->dy_pk = c.take<uint16_t>((size_t)Rp * m->dff);
        w->dh = c.take<float>((size_t)Rp * d);
        w->logits = c.take<float>((size_t)Rp * round_up(m->V, 32));
        w->ptop = c.take<float4>((size_t)Rp * (round_up(m->V, 32) / 32));
        w->stopv = c.take<float>((size_t)Rp * 4);
        {
            int ldmax = 3 * inner;
            if (m->dff > ldmax) ldmax = m->dff;
            if (d > ldmax) ldmax = d;
            w->slab_stride = (size_t)Rp * ldmax;
            w->slabs = c.take<float>(16 * w->slab_stride);
            w->tickets = c.take<int>(512);
            w->rs_part = c.take<float>((size_t)Rp * (d / 8));
            w->rs_part1 = c.take<float>((size_t)Rp * (d / 8));
            w->rs_part2 = c.take<float>((size_t)Rp * (d / 8));
            w->xa = c.take<uint16_t>((size_t)Rp * (d + inner));
            w->xb = c.take<uint16_t>((size_t)Rp * (d + inner));
        }
        w->next_ids = c.take<int64_t>(Rp);
        w->unfinished = c.take<int>(Rp);
        w->anc = c.take<int>((size_t)m->T_cap * R);
        w->beam_idx = c.take<int>(Rp);
        w->beam_div = c.take<float>((size_t)m->T_cap + 1);
        w->beam_state = c.take<char>(K > 1 ? beam_state_bytes(B, K, max_len) : 16);
    }
    if (T > 0) {
        const int T_cap = round_up(T, 64);
        const size_t MT = (size_t)B * T_cap;
        w->tf_ids = c.take<int64_t>(MT);
        w->tf_mask = c.take<uint8_t>(MT);
        w->tf_rowmap = c.take<int>(MT);
        w->tf_hidden = c.take<float>(MT * d);
        w->tf_x = c.take<uint16_t>(MT * d);
        w->tf_q = c.take<uint16_t>(MT * inner);
        w->tf_k = c.take<uint16_t>(MT * inner);
        w->tf_vt = c.take<uint16_t>(MT * inner);
        w->tf_ctx = c.take<uint16_t>(MT * inner);
        w->tf_y = c.take<uint16_t>(MT * m->dff);
        w->tf_xk = c.take<uint16_t>((size_t)B * Sx_cap * inner);
        w->tf_xvt = c.take<uint16_t>((size_t)B * Sx_cap * inner);
        w->tf_xc = c.take<uint16_t>((size_t)round_up(B * T, 32) * d);
    }
    w->total = align_up(c.off, 256);
}

// ---- continuous decoding (mg_generate_stream) ---------------------------------------------------------------------------
// stream counters (device, `ctr`): [0] live slots after the last step, [1] images finished, [2] steps run, [3] unused,
// [4] queue head (next image to hand to a slot), [5] images whose cross K/V are in the pool, [7] oldest live image
// (every image below it has finished: its pool entry may be overwritten), [8] N
struct StreamWs {
    Ws enc;                   // encoder workspace of one chunk
    uint16_t *xk, *xv;        // K/V pool [layer][pool entry][H][Sx_cap][64]  (absorbed form: null; encx = pool of encoder states [pool entry][Sx_cap][d])
    uint16_t *encx, *qx, *xpart;
    float *xml;
    size_t pool_stride;       // elements between layers
    int* xlen_pool;           // [pool entries]
    uint16_t *sk, *sv, *dq, *dx_pk, *dy_pk, *xa, *xb;
    float *dh, *logits, *rs_part, *rs_part1, *rs_part2, *kpart;
    int* tickets;
    int64_t* next_ids;
    int *unfinished, *pos, *img, *pool, *ctr, *err;
    // beam queue form (K > 1): `slots` image slots of K rows each; per-slot K/V owner, slot -> image assignment of the step, ancestor
    // table, beam bookkeeping of the slots
    int *bpool, *assign, *anc, *beam_idx;
    float* beam_div;
    void* beam_state;
    size_t total;
};
void carve_stream(const mg_model* m, char* base, int chunk, int L, int slots_img, int pool_chunks, StreamWs* w, int K = 1, int max_len = 0) {
    carve(m, base, chunk, L, 1, 0, 0, m->e1_M, &w->enc);
    Carver c{base};
    c.off = w->enc.total;
    const int d = m->d, inner = m->inner, H = m->H;
    const int slots = slots_img * K;                        // decode rows
    const int Sx_cap = round_up(L + m->P, 64) + round_up(m->e1_M, 64), Rp = round_up(slots, 32);      // keys of an image: [e1 tokens | encoder positions]
    const size_t nl = m->dec.size(), entries = (size_t)pool_chunks * chunk;
    w->pool_stride = entries * H * Sx_cap * 64;
    w->xk = w->xv = w->encx = w->qx = w->xpart = nullptr; w->xml = nullptr;
    if (use_absorb(m, K, slots)) {
        w->encx = c.take<uint16_t>(entries * Sx_cap * d);
        w->qx = c.take<uint16_t>((size_t)Rp * H * d);
        w->xpart = c.take<uint16_t>((size_t)Rp * m->xa_split * H * d);
        w->xml = c.take<float>((size_t)Rp * m->xa_split * H * 2);
    } else {
        w->xk = c.take<uint16_t>(nl * w->pool_stride);
        w->xv = c.take<uint16_t>(nl * w->pool_stride);
    }
    w->xlen_pool = c.take<int>(entries);
    w->sk = c.take<uint16_t>(nl * slots * H * (size_t)m->T_cap * 64);
    w->sv = c.take<uint16_t>(nl * slots * H * (size_t)m->T_cap * 64);
    w->dq = c.take<uint16_t>((size_t)Rp * inner);
    w->dx_pk = c.take<uint16_t>((size_t)Rp * d);
    w->dy_pk = c.take<uint16_t>((size_t)Rp * m->dff);
    w->dh = c.take<float>((size_t)Rp * d);
    w->logits = c.take<float>((size_t)Rp * round_up(m->V, 32));
    w->rs_part = c.take<float>((size_t)Rp * (d / 8));
    w->rs_part1 = c.take<float>((size_t)Rp * (d / 8));
    w->rs_part2 = c.take<float>((size_t)Rp * (d / 8));
    w->xa = c.take<uint16_t>((size_t)Rp * (d + inner));
    w->xb = c.take<uint16_t>((size_t)Rp * (d + inner));
    {
        int ldmax = 3 * inner;
        if (m->dff > ldmax) ldmax = m->dff;
        if (d > ldmax) ldmax = d;
        w->kpart = c.take<float>((size_t)16 * Rp * ldmax);
        w->tickets = c.take<int>(512);
    }
    w->next_ids = c.take<int64_t>(Rp);
    w->unfinished = c.take<int>(Rp);
    w->pos = c.take<int>(Rp);
    w->img = c.take<int>(Rp);
    w->pool = c.take<int>(Rp);
    w->ctr = c.take<int>(64);
    w->err = c.take<int>(64);
    w->bpool = c.take<int>(round_up(slots_img, 32));
    w->assign = c.take<int>(round_up(slots_img, 32));
    if (K > 1) {
        w->anc = c.take<int>((size_t)m->T_cap * slots);
        w->beam_idx = c.take<int>(Rp);
        w->beam_div = c.take<float>((size_t)m->T_cap + 1);
        w->beam_state = c.take<char>(beam_state_bytes(slots_img, K, max_len));
    } else {
        w->anc = nullptr; w->beam_idx = nullptr; w->beam_div = nullptr; w->beam_state = nullptr;
    }
    w->total = align_up(c.off, 256);
}
__global__ __launch_bounds__(256) void stream_init_kernel(int64_t* out_ids, int* out_len, int N, int max_len, int64_t start, int64_t pad,
                                                     int* unfinished, int* pos, int* img, int* pool, int64_t* next_ids, int slots, int* ctr, int* err) {
    const size_t n = (size_t)N * max_len;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out_ids[i] = (i % max_len) == 0 ? start : pad;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < N; i += blockDim.x) out_len[i] = 0;
        for (int r = threadIdx.x; r < slots; r += blockDim.x) { unfinished[r] = 0; pos[r] = 0; img[r] = -1; pool[r] = 0; next_ids[r] = start; }
        if (threadIdx.x < 16) ctr[threadIdx.x] = threadIdx.x == 8 ? N : 0;
        if (threadIdx.x < 2) err[threadIdx.x] = 0;
    }
}
// a chunk's cross K/V are in the pool: publish its key counts and make its images available to the slots
// encoder stream, end of a chunk: its key counts go to the pool entries, its input-error count is accumulated
__global__ __launch_bounds__(64) void stream_chunk_done_kernel(const int* xlen_chunk, int* xlen_pool, int entry0, int n, const int* enc_counters, int* err) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) xlen_pool[entry0 + i] = xlen_chunk[i];
    if (threadIdx.x == 0) {
        err[0] += enc_counters[3];
        int keys = 0;
        for (int i = 0; i < n; ++i) keys += xlen_chunk[i];
        err[1] += keys;                    // attended positions over all images (statistics: K/V bytes a step streams)
    }
}
// decode stream, after the chunk's event: its images may be handed to slots
__global__ __launch_bounds__(64) void stream_ready_kernel(int n, int* ctr) {
    if (threadIdx.x == 0) ctr[5] += n;
}

// greedy calls stream the encoder states (k_xattn.hip); beam search keeps the per-layer K / V streams (its G rows of an image share one pass)
// (auto: from 96 decode rows on - below that the stream's one workgroup per row is latency-bound and the K / V form's 16 workgroups per row win)
bool use_absorb(const mg_model* m, int K, int rows) { return K == 1 && (m->absorb == 1 || (m->absorb == 2 && rows >= MG_ABSORB_AUTO_ROWS)); }

int check_launch(const char* what) {
    const int e = mg_peek_error();
    if (e != 0) {
        const MgErrSite site = mg_err_site();
        mg_err_site() = MgErrSite{0, nullptr};
        return fail(MG_E_HIP, "%s: HIP error %d (%s)%s%s", what, e, mg_error_string(e), site.what ? ", first failing call: " : "",
                    site.what ? site.what : "");
    }
    return MG_OK;
}
// One call at a time per execution context: a second thread entering the same context gets MG_E_STATE instead of racing on its buffers.
#define MG_ONE_CALL(m, who)                                                                                         \
    std::unique_lock<std::recursive_mutex> call_lock((m)->call_mu, std::try_to_lock);                               \
    if (!call_lock.owns_lock())                                                                                    \
        return fail(MG_E_STATE, who ": this execution context is inside another call (one call at a time per context; mg_clone gives further contexts)")
// Start of a compute entry point: the runtime's last-error slot is per host thread and shared with every other library in the process -
// whatever is pending there was not caused by this call.
// (not inside a nested call: mg_generate_stream runs mg_encode per chunk in the middle of its decode loop, and a launch failure of
// the steps enqueued before it must survive until the outer call's check_launch)
static thread_local int g_nested_entry = 0;
struct NestedEntry { NestedEntry() { ++g_nested_entry; } ~NestedEntry() { --g_nested_entry; } };
void entry_drain() {
    if (g_nested_entry > 0) return;
    (void)mg_peek_error();
    mg_err_site() = MgErrSite{0, nullptr};
}

GemmArgs gemm_args(const uint16_t* X, const uint16_t* W, int M, int N, int K) {
    GemmArgs a{};
    a.X = X; a.W = W; a.M = M; a.N = N; a.K = K;
    return a;
}
void set_heads(GemmArgs& a, int H, int S_in, int S_cap, uint16_t* p0, int f0, uint16_t* p1, int f1, uint16_t* p2, int f2) {
    a.heads.ptr[0] = p0; a.heads.ptr[1] = p1; a.heads.ptr[2] = p2;
    a.heads.fmt[0] = f0; a.heads.fmt[1] = f1; a.heads.fmt[2] = f2;
    a.heads.inner = H * 64; a.heads.H = H; a.heads.S_in = S_in; a.heads.S_cap = S_cap;
}

// one FFN sub-layer on M rows: hidden += wo2 · relu(wi · RMSNorm(hidden))
void ffn_block(const mg_model* m, bool rows_mode, float* hidden, uint16_t* x_pk, uint16_t* y_pk, int M, size_t ln, size_t wi,
               size_t wo2, mgStream_t st) {
    rmsnorm_pack(hidden, m->at<float>(ln), x_pk, nullptr, M, m->d, m->c.layer_norm_epsilon, 1.0f, st);
    GemmArgs a = gemm_args(x_pk, m->at<uint16_t>(wi), M, m->dff, m->d);
    a.out_pk = y_pk;
    rows_mode ? gemm_rows(a, EPI_PK_RELU, st) : gemm(a, EPI_PK_RELU, st);
    GemmArgs b = gemm_args(y_pk, m->at<uint16_t>(wo2), M, m->d, m->dff);
    b.out_f32 = hidden; b.ldo = m->d;
    rows_mode ? gemm_rows(b, EPI_F32_RESID, st) : gemm(b, EPI_F32_RESID, st);
}

}  // namespace

// Everything one decode step needs besides the model: the buffers of the live rows, the cross K/V streams they read and the
// selection state.  Two users: mg_generate (one batch, all rows at the same position) and mg_generate_stream (continuous
// decoding: `slots.pos` non-null, every row at its own position on its own image, K/V streams in a pool).
struct DecodeCtx {
    uint16_t *xk, *xv;        // cross K/V: [layer][owner][H][Sx_cap][64]
    // absorbed form (encx non-null): the attended encoder states [owner][Sx_cap][d] + the scratch of the three launches
    uint16_t *encx, *qx, *xpart;
    float *xml;
    size_t xkv_stride;        // elements between layers
    int Sx_cap;
    const int* xlen;          // keys per K/V owner
    uint16_t *sk, *sv;        // self-attention caches [layer][row][H][T_cap][64]
    size_t skv_stride;
    uint16_t *dq, *dx_pk, *dy_pk, *xa, *xb;
    float *dh, *logits, *rs_part, *rs_part1, *rs_part2;
    float* kpart;             // K-slab projections with several row tiles: partial sums [16][rows padded][<= ldmax], and their arrival counters
    int* tickets;
    float4* ptop;             // fused greedy tail (null: separate embed / selection launches)
    float* stopv;
    int64_t* next_ids;
    int *unfinished, *anc, *beam_idx;
    float* beam_div;
    void* beam_state;
    int* counters;
    int B, K, R, max_length, min_length, early_stopping;
    float length_penalty;
    int64_t* out_ids;
    float* step_top2;
    const int* live;          // rows skipped by the attention launches (finished / idle), nullable
    SlotTable slots;          // continuous decoding
    // continuous BEAM decoding (slots.pos non-null and K > 1): B image slots of K rows; per-slot K/V owner, this step's slot -> image
    // assignment, outputs of the images that stop
    int *bpool, *assign;
    int32_t* out_len;
    float* out_scores;
};

// Decode step, 6 launches per layer: QKV -> self-attention -> [O residual | cross-Q] -> cross-attention ->
// [cross-O residual | FFN wi] -> FFN wo residual.  Residual projections run with complete sums per workgroup and
// leave per-row partial sums of squares; RMSNorm being a per-row scalar, every consumer applies
// rsqrt(mean(h^2)+eps) itself (RowScale) — there are no norm launches.  The two bracketed pairs use the product
// weights built by mg_finalize: the second projection of a pair reads [bf16(h before the residual) | context]
// and so does not wait for the residual projection next to it.
// tdev == nullptr: step-dependent values are passed by value (eager launches); otherwise the kernels read the step from the
// device counter, which makes the launch sequence capturable as a graph.  Continuous decoding (c.slots.pos): positions,
// images and K/V owners come from the device slot table, t / tdev are not used.
static void decode_step(mg_model* m, const DecodeCtx& c, int t, const int* tdev, bool time_cross, mgStream_t st) {
    const int d = m->d, H = m->H, inner = m->inner, T_cap = m->T_cap, R = c.R, K = c.K;
    const size_t nl = m->dec.size();
    const float eps = m->c.layer_norm_epsilon;
    const int ldl = round_up(m->V, 32);
    const int K2 = d + inner, kts2 = K2 >> 4, kt_ctx = d >> 4;
    const int64_t pad = m->c.pad_token_id;
    const bool stream = c.slots.pos != nullptr;
    int* counters = c.counters;
    const int max_length = c.max_length, min_length = c.min_length;
    const int* live = c.live;
    const size_t skv_stride = c.skv_stride, xkv_stride = c.xkv_stride;
    const int Sx_cap = c.Sx_cap, B = c.B;
    int64_t* out_ids = c.out_ids;
    float* step_top2 = c.step_top2;
    const float length_penalty = c.length_penalty;
    const int early_stopping = c.early_stopping;
    RowScale none{};
    RowScale rs0{c.rs_part, d / 8, 1.0f / (float)d, eps};     // after the FFN output (next layer's ln0 / final norm)
    RowScale rs1{c.rs_part1, d / 8, 1.0f / (float)d, eps};    // after the self-attention output (cross-attention norm)
    RowScale rs2{c.rs_part2, d / 8, 1.0f / (float)d, eps};    // after the cross-attention output (FFN norm)
    // Fused tail (greedy batch calls): the selection kernel of step t has already left the embedding + first norm of step t + 1
    // (the caller embeds the start token once in front of the first step); otherwise a step starts with its own embedding launch.
    const bool fused_tail = c.ptop != nullptr;
    // tools build only (tools/whatif_decode.py: WRONG results, valid timing): launches of the step left out by bit - 1 QKV, 2 self-attention,
    // 4 [Wo | cross-Q], 8 cross-attention, 16 [Wxo | FFN-wi], 32 FFN-wo, 64 lm_head
#ifdef MG_TOOLS
    static const int whatif = [] { const char* e = getenv("MG_WHATIF_STEP"); return e ? atoi(e) : 0; }();
#else
    constexpr int whatif = 0;
#endif
    if (!fused_tail)
        embed_norm_rows(c.next_ids, m->at<uint16_t>(m->tok_emb), c.dh, m->at<float>(m->dec[0].ln0), c.dx_pk, c.xa, K2, 0, R, d, m->V,
                        counters + 3, eps, st);
    for (size_t li = 0; li < nl; ++li) {
        const DecLayer& l = m->dec[li];
        uint16_t* sk = c.sk + li * skv_stride;
        uint16_t* sv = c.sv + li * skv_stride;
        {
            {
                GemmArgs a = gemm_args(c.dx_pk, m->at<uint16_t>(l.wqkv), R, 3 * inner, d);
                set_heads(a, H, R, T_cap, c.dq, HF_STEP_Q, sk, HF_STEP_KV, sv, HF_STEP_KV);
                a.heads.pos = t; a.heads.pos_dev = tdev; a.heads.pos_rows = c.slots.pos;
                a.rs = li == 0 ? none : rs0;      // layer 0 reads the explicitly normalised embedding
                a.both_halves = m->shared_gpu;    // (other contexts beside this one: fewer activation re-reads through L2 beat more workgroups)
                if (!(whatif & 1)) gemm_rows(a, EPI_HEADS, st);      // q -> dq, k/v appended to the cache at position t
            }
            AttnStepArgs s{};
            s.q = c.dq; s.Kc = sk; s.Vc = sv; s.ctx = c.xa; s.ctx_ld = K2; s.ctx_col0 = d; s.rows = R; s.H = H; s.group = 1;
            s.cap = T_cap; s.n_keys = t + 1; s.bias = m->at<float>(m->dec_tab); s.anc = K > 1 ? c.anc : nullptr; s.t = t; s.t_dev = tdev;
            s.live = live; s.pos_rows = c.slots.pos;
            if (!(whatif & 2)) attention_step(s, st);
        }
        {   // h += Wo·ctx (partials of sum h^2 -> rs1, bf16(h) -> xb)   |   cross-attention q (un-normalised) -> dq
            ResidArgs r{};
            r.X = c.xa; r.x_kts = kts2; r.x_k0 = kt_ctx; r.W = m->at<uint16_t>(l.wo); r.h = c.dh; r.x2_pk = c.xb; r.x2_ld = K2;
            r.part = c.rs_part1; r.M = R; r.N = d; r.K = inner;
            GemmArgs g = gemm_args(c.xa, m->at<uint16_t>(l.xq2), R, inner, K2);
            set_heads(g, H, R, T_cap, c.dq, HF_STEP_Q, nullptr, HF_NONE, nullptr, HF_NONE);
            g.both_halves = m->shared_gpu;
            if (!(whatif & 4)) gemm_rows_pair(r, g, EPI_HEADS, st);
        }
        // cross-attention over the image's compacted K/V stream (all beams of an image share one pass)
        AttnStepArgs x{};
        x.q = c.dq; x.qrs = rs1; x.Kc = c.xk + li * xkv_stride; x.Vc = c.xv + li * xkv_stride; x.ctx = c.xb; x.ctx_ld = K2;
        x.ctx_col0 = d; x.rows = R; x.H = H; x.group = K; x.cap = Sx_cap; x.len = c.xlen;
        x.live = live; x.kv_owner = (K > 1 && stream) ? c.bpool : c.slots.pool;
        x.one_wg_per_cu = m->shared_gpu;       // (beams: one owner per image slot = group of K rows)
        const bool timed = time_cross && m->prof_used + 3 <= m->prof_ev.size();
        XAttnArgs xa{};
        if (c.encx) {      // weight-absorbed form: q' = q·Wk_h | stream of the states | ctx_h = c_h·Wv_h^T   (the bracket times the stream)
            xa.q = c.dq; xa.qx = c.qx; xa.wk = m->at<uint16_t>(l.xwk); xa.wv = m->at<uint16_t>(l.xwv); xa.enc = c.encx; xa.len = c.xlen;
            xa.kv_owner = c.slots.pool; xa.live = live; xa.qrs = rs1; xa.part = c.xpart; xa.ml = c.xml; xa.ctx = c.xb; xa.ctx_ld = K2; xa.ctx_col0 = d;
            xa.rows = R; xa.H = H; xa.d = d; xa.cap = Sx_cap; xa.nsplit = m->xa_split; xa.nstg = m->xa_stages; xa.nt = m->xa_nt;
            if (!(whatif & 8)) xattn_expand(xa, st);
        }
        if (timed) mg_event_record(m->prof_ev[m->prof_used], st);
        if (!(whatif & 8)) { if (c.encx) xattn_stream(xa, st); else attention_step(x, st); }
        if (timed) {   // third event right behind the second: the empty bracket calibrates what two records alone cost
            mg_event_record(m->prof_ev[m->prof_used + 1], st);
            mg_event_record(m->prof_ev[m->prof_used + 2], st);
            m->prof_used += 3;
        }
        if (c.encx && !(whatif & 8)) xattn_contract(xa, st);
        {   // h += Wxo·ctx_x (partials -> rs2)   |   y = relu(wi·...) un-normalised -> dy_pk
            ResidArgs r{};
            r.X = c.xb; r.x_kts = kts2; r.x_k0 = kt_ctx; r.W = m->at<uint16_t>(l.xo); r.h = c.dh; r.part = c.rs_part2;
            r.M = R; r.N = d; r.K = inner;
            GemmArgs g = gemm_args(c.xb, m->at<uint16_t>(l.wi2), R, m->dff, K2);
            g.out_pk = c.dy_pk;
            if (!(whatif & 16)) gemm_rows_pair(r, g, EPI_PK_RELU, st);
        }
        {   // FFN output (input scaled by rs2); leaves bf16(h·gain) for the next QKV / lm_head (gain = next layer's ln0,
            // or the final norm with the d_model^-0.5 of the tied head), bf16(h) for the next pair, partials -> rs0
            const bool last = li + 1 == nl;
            ResidArgs r{};
            r.X = c.dy_pk; r.W = m->at<uint16_t>(l.wo2); r.h = c.dh; r.gain = m->at<float>(last ? m->dec_ln : m->dec[li + 1].ln0);
            r.gscale = (last && m->tied) ? 1.0f / sqrtf((float)d) : 1.0f; r.x_pk = c.dx_pk; r.x2_pk = c.xa; r.x2_ld = K2; r.part = c.rs_part;
            r.M = R; r.N = d; r.K = m->dff; r.rs = rs2;
            r.wide_tiles = 8;      // the same K partition (16 waves) for every call (up to 256 rows: 8 greedy batches of 32; beam-5 at batch 32 = 5 tiles)
            r.kpart = c.kpart; r.ticket = c.tickets; r.alone = m->shared_gpu ? 0 : 1;
            if (!(whatif & 32)) gemm_rows_resid(r, st);
        }
    }
    if (fused_tail) {          // lm_head with per-workgroup top-2 partials (no fp32 logits), stop token kept apart for MinLength
        TopOut top{c.ptop, c.stopv, {m->c.eos_token_id, -1, -1, -1}, 0};
        if (!(whatif & 64)) gemm_rows_splitk(c.dx_pk, m->at<uint16_t>(m->lm_head), c.logits, R, m->V, d, ldl, 0, 1, rs0, st, &top);
    } else {
        gemm_rows_splitk(c.dx_pk, m->at<uint16_t>(m->lm_head), c.logits, R, m->V, d, ldl, 0, 1, rs0, st);
    }
    if (m->dbg_logits && t < m->dbg_steps)
        MG_LAUNCH(capture_logits_kernel, dim3(1024), dim3(256), 0, st, (const float*)c.logits, ldl,
                  m->dbg_logits + (size_t)t * R * m->V, R, m->V);
    if (K == 1) {
        ArgmaxArgs g{};
        g.logits = c.logits; g.rows = R; g.V = m->V; g.ldl = ldl; g.eos = m->c.eos_token_id; g.pad = (int)pad;
        g.min_len = min_length; g.next_ids = c.next_ids; g.out_ids = out_ids; g.max_len = max_length;
        g.pos = tdev ? 1 : t + 1; g.pos_dev = tdev;
        g.unfinished = c.unfinished; g.n_unfinished = counters + 5;
        g.top2 = step_top2 ? (tdev ? step_top2 : step_top2 + (size_t)(t + 1) * R * 2) : nullptr;
        g.step_ctr = stream ? nullptr : counters;      // the last workgroup to finish does the step bookkeeping (no step_end launch)
        g.slots = c.slots;
        if (fused_tail) {
            g.ptop = c.ptop; g.stopv = c.stopv; g.ntiles = ldl / 32;
            g.tok_emb = m->at<uint16_t>(m->tok_emb); g.h = c.dh; g.gain = m->at<float>(m->dec[0].ln0); g.x_pk = c.dx_pk;
            g.x2_pk = c.xa; g.x2_ld = K2; g.x2_col0 = 0; g.d = d; g.eps = eps;
            greedy_select_fused(g, st);
        } else {
            greedy_select(g, st);
        }
        if (stream) slot_refill(c.slots, c.next_ids, c.unfinished, R, st);
        if (m->dbg_forced && t + 1 < max_length)
            MG_LAUNCH(force_ids_kernel, dim3((R + 63) / 64), dim3(64), 0, st, c.next_ids, m->dbg_forced, R, max_length, t + 1);
    } else if (stream) {
        // queue form: every slot at its own length; stopped images are written out and their slots handed to the next images
        const BeamSlots bs{c.slots.pos, c.unfinished};
        beam_step(c.beam_state, c.logits, ldl, m->V, B, K, max_length, 0, nullptr, c.beam_div, m->c.eos_token_id, min_length,
                  length_penalty, early_stopping, c.next_ids, c.beam_idx, counters, st, &bs);
        beam_reorder_anc(c.anc, c.beam_idx, R, max_length - 1, nullptr, counters, st, &bs);
        beam_slots_step(c.beam_state, B, K, max_length, (int)pad, m->c.eos_token_id, c.slots.start_id, early_stopping, c.slots.pos, c.slots.img,
                        c.slots.pool, c.bpool, c.unfinished, c.assign, c.next_ids, c.anc, T_cap, c.slots.pool_cap, out_ids, c.out_len,
                        c.out_scores, c.slots.ctr, true, st);
    } else {
        beam_step(c.beam_state, c.logits, ldl, m->V, B, K, max_length, t + 1, tdev, c.beam_div, m->c.eos_token_id, min_length,
                  length_penalty, early_stopping, c.next_ids, c.beam_idx, counters, st);
        beam_reorder_anc(c.anc, c.beam_idx, R, tdev ? max_length - 1 : t + 1, tdev, counters, st);
    }
    if (K > 1 && !stream) MG_LAUNCH(step_end_kernel, dim3(1), dim3(64), 0, st, counters, 0);
}

extern "C" {

const char* mg_last_error(void) { return g_err.c_str(); }

int mg_create(const mg_config* cfg, mg_model** out) {
    if (!cfg || !out) return fail(MG_E_ARG, "mg_create: null argument");
    const mg_config& c = *cfg;
    if (c.d_kv != 64) return fail(MG_E_UNSUPPORTED, "d_kv must be 64 (got %d)", c.d_kv);
    if (c.d_model % 64 || c.d_ff % 64) return fail(MG_E_UNSUPPORTED, "d_model and d_ff must be multiples of 64");
    if (c.image_size % c.patch_size || c.patch_size % 8) return fail(MG_E_UNSUPPORTED, "bad image/patch size");
    const int Kpatch = c.num_channels * c.patch_size * c.patch_size;
    if (Kpatch % 64) return fail(MG_E_UNSUPPORTED, "channels*patch^2 must be a multiple of 64");
    if (c.num_heads < 1 || c.num_layers < 1 || c.num_decoder_layers < 1 || c.vocab_size < 2)
        return fail(MG_E_ARG, "bad config");
    mg_model* m = new mg_model();
    m->c = c;
    m->d = c.d_model; m->H = c.num_heads; m->inner = c.num_heads * 64; m->dff = c.d_ff; m->V = c.vocab_size;
    m->n_side = c.image_size / c.patch_size; m->P = m->n_side * m->n_side; m->Kpatch = Kpatch;
    m->M2 = c.max_2d_position_embeddings;
    m->T_cap = round_up(c.max_decode_len > 0 ? c.max_decode_len : 512, 64);
    m->tied = c.tie_word_embeddings != 0;
    // The runtime's queue-thread mode (AMD_DIRECT_DISPATCH=0, a debugging switch) replays captured decode steps wrongly on this ROCm
    // (all-pad ids, tests/test_engine.py graph-mode test under that switch): launch eagerly there.
    { const char* e = getenv("AMD_DIRECT_DISPATCH"); if (e && e[0] == '0') m->use_graph = 0; }
    { const char* e = getenv("MG_ENC_ROW_TILES"); if (e && e[0] == '0') m->row_tiles = false; }
    { const char* e = getenv("MG_DECODE_FUSED_TAIL"); if (e && e[0] == '0') m->fused_tail = false; }
    { const char* e = getenv("MG_XATTN_ABSORB"); if (e && e[0] >= '0' && e[0] <= '2') m->absorb = e[0] - '0'; }
    { const char* e = getenv("MG_PACE"); if (e && e[0] == '0') m->pace = 0; }
    { const char* e = getenv("MG_XATTN_NT"); if (e && e[0] == '0') m->xa_nt = 0; }
    { const char* e = getenv("MG_XATTN_SPLIT"); if (e && atoi(e) >= 1 && atoi(e) <= 4) m->xa_split = atoi(e); }
    { const char* e = getenv("MG_XATTN_STAGES"); if (e && (atoi(e) == 3 || atoi(e) == 4)) m->xa_stages = atoi(e); }
    if (!xattn_supported(c.d_model, c.num_heads)) m->absorb = 0;
    // arena layout
    size_t off = 0;
    auto take = [&](size_t bytes) { off = align_up(off, 256); size_t o = off; off += bytes; return o; };
    const int d = m->d, inner = m->inner, dff = m->dff, H = m->H;
    m->tok_emb = take((size_t)m->V * d * 2);
    m->lm_head = take(pk_elems(m->V, d) * 2);
    m->patch_w = take(pk_elems(d, Kpatch) * 2);
    m->patch_b = take((size_t)d * 4);
    m->x_emb = take((size_t)m->M2 * d * 2);
    m->y_emb = take((size_t)m->M2 * d * 2);
    const int nb = c.relative_attention_num_buckets;
    for (int i = 0; i < 3; ++i) m->rb_raw[i] = take((size_t)nb * H * 4);
    m->rb_dec_raw = take((size_t)nb * H * 4);
    m->dec_tab = take((size_t)m->T_cap * H * 4);
    m->bk1 = take(257 * 4); m->bkhv = take(201 * 4); m->bkdec = take((size_t)m->T_cap * 4);
    m->enc_ln = take((size_t)d * 4); m->dec_ln = take((size_t)d * 4);
    for (int i = 0; i < c.num_layers; ++i) {
        EncLayer l;
        l.wqkv = take(pk_elems(3 * inner, d) * 2); l.wo = take(pk_elems(d, inner) * 2); l.ln0 = take((size_t)d * 4);
        l.wi = take(pk_elems(dff, d) * 2); l.wo2 = take(pk_elems(d, dff) * 2); l.ln1 = take((size_t)d * 4);
        m->enc.push_back(l);
    }
    for (int i = 0; i < c.num_decoder_layers; ++i) {
        DecLayer l;
        l.wqkv = take(pk_elems(3 * inner, d) * 2); l.wo = take(pk_elems(d, inner) * 2); l.ln0 = take((size_t)d * 4);
        l.xq = take(pk_elems(inner, d) * 2); l.xkv = take(pk_elems(2 * inner, d) * 2); l.xo = take(pk_elems(d, inner) * 2);
        l.ln1 = take((size_t)d * 4);
        l.wi = take(pk_elems(dff, d) * 2); l.wo2 = take(pk_elems(d, dff) * 2); l.ln2 = take((size_t)d * 4);
        l.xq2 = take(pk_elems(inner, d + inner) * 2); l.wi2 = take(pk_elems(dff, d + inner) * 2);
        l.xwk = take((size_t)H * d * 64 * 2); l.xwv = take((size_t)H * d * 64 * 2);
        m->dec.push_back(l);
    }
    {   // fp32 scratch of mg_finalize (unpacked factors and one product)
        const size_t nmax = (size_t)(dff > 2 * inner ? dff : 2 * inner);
        m->fin_a = take(nmax * d * 4); m->fin_b = take((size_t)d * inner * 4); m->fin_c = take(nmax * (d + inner) * 4);
    }
    m->arena_bytes = align_up(off, 256);
    // host bucket tables
    const int md = c.relative_attention_max_distance;
    for (int i = 0; i < 257; ++i) m->h_bk1.push_back(bucket_of(i - 128, true, nb, 128));
    for (int i = 0; i < 201; ++i) m->h_bkhv.push_back(bucket_of(i - 100, true, nb, 100));
    for (int i = 0; i < m->T_cap; ++i) m->h_bkdec.push_back(bucket_of(-(long)i, false, nb, md));
    *out = m;
    return MG_OK;
}

void mg_destroy(mg_model* m) { delete m; }

int mg_clone(const mg_model* src, mg_model** out) {
    if (!src || !out) return fail(MG_E_ARG, "mg_clone: null argument");
    if (!src->arena || !src->finalized) return fail(MG_E_STATE, "mg_clone: the source model has no finalized weights");
    mg_model* m = new mg_model();
    // geometry, arena layout and switches are copied; everything a call mutates (graphs, events, streams, profiling) starts fresh
    m->c = src->c;
    m->d = src->d; m->H = src->H; m->inner = src->inner; m->dff = src->dff; m->V = src->V; m->P = src->P; m->n_side = src->n_side;
    m->Kpatch = src->Kpatch; m->M2 = src->M2; m->T_cap = src->T_cap;
    m->arena = src->arena; m->arena_bytes = src->arena_bytes;
    m->tok_emb = src->tok_emb; m->lm_head = src->lm_head; m->patch_w = src->patch_w; m->patch_b = src->patch_b;
    m->x_emb = src->x_emb; m->y_emb = src->y_emb;
    for (int i = 0; i < 3; ++i) m->rb_raw[i] = src->rb_raw[i];
    m->rb_dec_raw = src->rb_dec_raw; m->dec_tab = src->dec_tab;
    m->bk1 = src->bk1; m->bkhv = src->bkhv; m->bkdec = src->bkdec; m->enc_ln = src->enc_ln; m->dec_ln = src->dec_ln;
    m->enc = src->enc; m->dec = src->dec; m->loaded = src->loaded; m->lm_head_loaded = src->lm_head_loaded; m->finalized = true;
    m->h_bk1 = src->h_bk1; m->h_bkhv = src->h_bkhv; m->h_bkdec = src->h_bkdec;
    m->fin_a = src->fin_a; m->fin_b = src->fin_b; m->fin_c = src->fin_c;
    m->use_graph = src->use_graph; m->enc_mode = src->enc_mode; m->enc_mask = src->enc_mask;
    m->row_tiles = src->row_tiles; m->trim_padding = src->trim_padding; m->fused_tail = src->fused_tail; m->tied = src->tied;
    m->absorb = src->absorb; m->xa_split = src->xa_split; m->xa_stages = src->xa_stages; m->pace = src->pace; m->xa_nt = src->xa_nt;
    m->e1m = src->e1m; m->e1_M = src->e1_M;
    *out = m;
    return MG_OK;
}

size_t mg_weights_bytes(const mg_model* m) { return m ? m->arena_bytes : 0; }

int mg_bind_weights(mg_model* m, void* arena) {
    if (!m || !arena) return fail(MG_E_ARG, "mg_bind_weights: null argument");
    m->arena = (char*)arena;
    m->loaded.clear();
    m->lm_head_loaded = false;
    m->finalized = false;
    return MG_OK;
}

// host copies of the bucket tables the bias tables are built from (parity tests pin them on the golden tables)
int mg_debug_bucket_table(const mg_model* m, int which, int* out_host, int n) {
    const std::vector<int>& v = which == 0 ? m->h_bk1 : (which == 1 ? m->h_bkhv : m->h_bkdec);
    if (n > (int)v.size()) n = (int)v.size();
    for (int i = 0; i < n; ++i) out_host[i] = v[i];
    return n;
}

int mg_load_tensor(mg_model* m, void* stream, const char* hf_key, const void* src, int dtype, const int64_t* shape,
                   int ndim) {
    if (!m || !hf_key || !src || !shape) return fail(MG_E_ARG, "mg_load_tensor: null argument");
    if (!m->arena) return fail(MG_E_STATE, "mg_load_tensor: bind a weights arena first");
    if (dtype != MG_F32 && dtype != MG_BF16) return fail(MG_E_ARG, "dtype must be MG_F32 or MG_BF16");
    mgStream_t st = (mgStream_t)stream;
    const std::string key(hf_key);
    const int d = m->d, inner = m->inner, dff = m->dff, H = m->H, nb = m->c.relative_attention_num_buckets;
    auto want = [&](std::initializer_list<int64_t> dims) {
        if ((int)dims.size() != ndim) return false;
        int i = 0;
        for (int64_t v : dims) if (shape[i++] != v) return false;
        return true;
    };
    auto bad_shape = [&]() { return fail(MG_E_SHAPE, "mg_load_tensor: unexpected shape for %s", hf_key); };
    auto packw = [&](size_t off, int row0, int N, int K) {   // rows [row0, row0+N) of a packed [*,K] matrix
        pack_weight(src, dtype, N, K, m->at<uint16_t>(off) + pk_elems(row0, K), round_up(N, 32), st);
    };
    auto vec32 = [&](size_t off, size_t n) { convert_to_f32(src, dtype, m->at<float>(off), n, st); };
    auto mark = [&](const std::string& canon) { m->loaded[canon] = true; m->finalized = false; return check_launch(hf_key); };

    if (key == "shared.weight" || key == "encoder.embed_tokens.weight" || key == "decoder.embed_tokens.weight") {
        if (!want({m->V, d})) return bad_shape();
        convert_to_bf16(src, dtype, m->at<uint16_t>(m->tok_emb), (size_t)m->V * d, st);
        if (m->tied) packw(m->lm_head, 0, m->V, d);               // tied head (stock:1405-1413)
        return mark("shared.weight");
    }
    if (key == "lm_head.weight") {
        if (!want({m->V, d})) return bad_shape();
        // tie_word_embeddings: the checkpoint's lm_head.weight is discarded and the head re-tied to shared.weight, as
        // HF's tie_weights() does after loading; only an untied config uses it (then without the d_model^-0.5 scale)
        if (m->tied) return MG_KEY_IGNORED;
        packw(m->lm_head, 0, m->V, d);
        m->lm_head_loaded = true;
        return mark("lm_head.weight");
    }
    if (key == "patch_embed.proj.weight" || key == "encoder.embed_patches.proj.weight") {
        if (!want({d, m->c.num_channels, m->c.patch_size, m->c.patch_size})) return bad_shape();
        packw(m->patch_w, 0, d, m->Kpatch);
        return mark("patch_embed.proj.weight");
    }
    if (key == "patch_embed.proj.bias" || key == "encoder.embed_patches.proj.bias") {
        if (!want({d})) return bad_shape();
        vec32(m->patch_b, d);
        return mark("patch_embed.proj.bias");
    }
    if (key.rfind("decoder.embed_patches.", 0) == 0 || key.rfind("decoder.relative_bias.", 0) == 0)
        return MG_KEY_IGNORED;   // present in UDOP state dicts, never used by the decoder (stock:1212-1213)
    if (key.rfind("encoder.molscribe_", 0) == 0) return MG_KEY_IGNORED;   // OCSR e1 branch: SURVEY.md §8 a7 ("next" f-2)
    if (key == "encoder.cell_2d_embedding.x_position_embeddings.weight" ||
        key == "encoder.cell_2d_embedding.y_position_embeddings.weight") {
        if (!want({m->M2, d})) return bad_shape();
        const bool isx = key.find(".x_position") != std::string::npos;
        convert_to_bf16(src, dtype, m->at<uint16_t>(isx ? m->x_emb : m->y_emb), (size_t)m->M2 * d, st);
        return mark(key);
    }
    if (key == "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight" ||
        key == "encoder.relative_bias.biases.0.relative_attention_bias.weight") {
        if (!want({nb, H})) return bad_shape();
        vec32(m->rb_raw[0], (size_t)nb * H);
        return mark("encoder.relative_bias.biases.0.relative_attention_bias.weight");
    }
    if (key == "encoder.relative_bias.biases.1.relative_attention_bias.weight" ||
        key == "encoder.relative_bias.biases.2.relative_attention_bias.weight") {
        if (!want({nb, H})) return bad_shape();
        vec32(m->rb_raw[key[29] - '0'], (size_t)nb * H);
        return mark(key);
    }
    if (key == "decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight") {
        if (!want({nb, H})) return bad_shape();
        vec32(m->rb_dec_raw, (size_t)nb * H);
        return mark(key);
    }
    if (key == "encoder.final_layer_norm.weight" || key == "decoder.final_layer_norm.weight") {
        if (!want({d})) return bad_shape();
        vec32(key[0] == 'e' ? m->enc_ln : m->dec_ln, d);
        return mark(key);
    }
    int li = -1, sub = -1;
    char rest[96] = {0};
    const bool is_enc = sscanf(hf_key, "encoder.block.%d.layer.%d.%95s", &li, &sub, rest) == 3;
    const bool is_dec = !is_enc && sscanf(hf_key, "decoder.block.%d.layer.%d.%95s", &li, &sub, rest) == 3;
    if (is_enc || is_dec) {
        const std::string r(rest);
        if (li < 0 || li >= (int)(is_enc ? m->enc.size() : m->dec.size())) return fail(MG_E_KEY, "layer index out of range: %s", hf_key);
        if (r == "layer_norm.weight") {
            if (!want({d})) return bad_shape();
            size_t off;
            if (is_enc) { if (sub > 1) return fail(MG_E_KEY, "unknown key %s", hf_key); off = sub == 0 ? m->enc[li].ln0 : m->enc[li].ln1; }
            else { if (sub > 2) return fail(MG_E_KEY, "unknown key %s", hf_key); off = sub == 0 ? m->dec[li].ln0 : (sub == 1 ? m->dec[li].ln1 : m->dec[li].ln2); }
            vec32(off, d);
            return mark(key);
        }
        const int ffn_sub = is_enc ? 1 : 2;
        if (sub == ffn_sub && (r == "DenseReluDense.wi.weight" || r == "DenseReluDense.wo.weight")) {
            const bool wi = r[15] == 'w' && r[16] == 'i';
            if (wi) { if (!want({dff, d})) return bad_shape(); packw(is_enc ? m->enc[li].wi : m->dec[li].wi, 0, dff, d); }
            else { if (!want({d, dff})) return bad_shape(); packw(is_enc ? m->enc[li].wo2 : m->dec[li].wo2, 0, d, dff); }
            return mark(key);
        }
        const bool self_att = sub == 0 && r.rfind("SelfAttention.", 0) == 0 && r.size() == 22 && r.substr(15) == ".weight";
        const bool cross_att = is_dec && sub == 1 && r.rfind("EncDecAttention.", 0) == 0 && r.size() == 24 && r.substr(17) == ".weight";
        if (self_att || cross_att) {
            const char p = self_att ? r[14] : r[16];
            if (p == 'o') {
                if (!want({d, inner})) return bad_shape();
                packw(self_att ? (is_enc ? m->enc[li].wo : m->dec[li].wo) : m->dec[li].xo, 0, d, inner);
                return mark(key);
            }
            if (p != 'q' && p != 'k' && p != 'v') return fail(MG_E_KEY, "unknown key %s", hf_key);
            if (!want({inner, d})) return bad_shape();
            if (self_att) packw(is_enc ? m->enc[li].wqkv : m->dec[li].wqkv, (p == 'q' ? 0 : (p == 'k' ? 1 : 2)) * inner, inner, d);
            else if (p == 'q') packw(m->dec[li].xq, 0, inner, d);
            else packw(m->dec[li].xkv, (p == 'k' ? 0 : 1) * inner, inner, d);
            return mark(key);
        }
    }
    return fail(MG_E_KEY, "mg_load_tensor: unknown key %s", hf_key);
}

int mg_finalize(mg_model* m, void* stream) {
    if (!m || !m->arena) return fail(MG_E_STATE, "mg_finalize: no weights bound");
    mgStream_t st = (mgStream_t)stream;
    std::vector<std::string> need = {
        "shared.weight", "patch_embed.proj.weight", "patch_embed.proj.bias",
        "encoder.cell_2d_embedding.x_position_embeddings.weight", "encoder.cell_2d_embedding.y_position_embeddings.weight",
        "encoder.relative_bias.biases.0.relative_attention_bias.weight",
        "encoder.relative_bias.biases.1.relative_attention_bias.weight",
        "encoder.relative_bias.biases.2.relative_attention_bias.weight",
        "decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight",
        "encoder.final_layer_norm.weight", "decoder.final_layer_norm.weight"};
    char buf[160];
    for (size_t i = 0; i < m->enc.size(); ++i) {
        for (const char* s : {"0.SelfAttention.q.weight", "0.SelfAttention.k.weight", "0.SelfAttention.v.weight",
                              "0.SelfAttention.o.weight", "0.layer_norm.weight", "1.DenseReluDense.wi.weight",
                              "1.DenseReluDense.wo.weight", "1.layer_norm.weight"}) {
            snprintf(buf, sizeof buf, "encoder.block.%zu.layer.%s", i, s);
            need.push_back(buf);
        }
    }
    for (size_t i = 0; i < m->dec.size(); ++i) {
        for (const char* s : {"0.SelfAttention.q.weight", "0.SelfAttention.k.weight", "0.SelfAttention.v.weight",
                              "0.SelfAttention.o.weight", "0.layer_norm.weight", "1.EncDecAttention.q.weight",
                              "1.EncDecAttention.k.weight", "1.EncDecAttention.v.weight", "1.EncDecAttention.o.weight",
                              "1.layer_norm.weight", "2.DenseReluDense.wi.weight", "2.DenseReluDense.wo.weight",
                              "2.layer_norm.weight"}) {
            snprintf(buf, sizeof buf, "decoder.block.%zu.layer.%s", i, s);
            need.push_back(buf);
        }
    }
    if (!m->tied) need.push_back("lm_head.weight");
    for (const std::string& k : need)
        if (!m->loaded.count(k)) return fail(MG_E_STATE, "mg_finalize: tensor %s was never loaded", k.c_str());
    const int H = m->H;
    mg_memcpy_async(m->at<int>(m->bk1), m->h_bk1.data(), 257 * 4, st);
    mg_memcpy_async(m->at<int>(m->bkhv), m->h_bkhv.data(), 201 * 4, st);
    mg_memcpy_async(m->at<int>(m->bkdec), m->h_bkdec.data(), (size_t)m->T_cap * 4, st);
    MG_LAUNCH(build_table_kernel, dim3(8), dim3(256), 0, st, (const float*)m->at<float>(m->rb_dec_raw), (const int*)m->at<int>(m->bkdec), m->at<float>(m->dec_tab), m->T_cap, H);
    // Product weights of the decode step.  With h1 = h + Wo·ctx the cross-attention query is
    //   Wxq·G1·h1 = (Wxq·G1)·h + (Wxq·G1·Wo)·ctx          (G1 = diag of the cross-attention layer-norm gain, stock:611-640)
    // and likewise the FFN input projection over h2 = h1 + Wxo·ctx_x is (Wi·G2)·h1 + (Wi·G2·Wxo)·ctx_x (stock:313-325),
    // both up to the per-row RMSNorm scalar, which the consumers apply (RowScale).  One GEMM over [bf16(h) | ctx] with
    // the concatenated weight then runs NEXT TO the residual projection instead of after it: 6 launches per decoder
    // layer instead of 8.  Products are formed in fp32 from the bf16 weights and rounded to bf16 once.
    {
        const int d = m->d, inner = m->inner, dff = m->dff, K2 = d + inner;
        float *A = m->at<float>(m->fin_a), *Bm = m->at<float>(m->fin_b), *C = m->at<float>(m->fin_c);
        for (DecLayer& l : m->dec) {
            unpack_weight(m->at<uint16_t>(l.wo), Bm, d, inner, st);
            unpack_weight(m->at<uint16_t>(l.xq), A, inner, d, st);
            scale_cols_f32(A, m->at<float>(l.ln1), C, inner, d, K2, st);
            gemm_f32_scaled(A, m->at<float>(l.ln1), Bm, C + d, inner, d, inner, K2, st);
            pack_weight(C, 0, inner, K2, m->at<uint16_t>(l.xq2), round_up(inner, 32), st);
            unpack_weight(m->at<uint16_t>(l.xo), Bm, d, inner, st);
            unpack_weight(m->at<uint16_t>(l.wi), A, dff, d, st);
            scale_cols_f32(A, m->at<float>(l.ln2), C, dff, d, K2, st);
            gemm_f32_scaled(A, m->at<float>(l.ln2), Bm, C + d, dff, d, inner, K2, st);
            pack_weight(C, 0, dff, K2, m->at<uint16_t>(l.wi2), round_up(dff, 32), st);
            // absorbed cross-attention weights (exact re-orderings of the bf16 K / V weights)
            if (xattn_supported(d, m->H)) {
                unpack_weight(m->at<uint16_t>(l.xkv), A, 2 * inner, d, st);
                xattn_pack_weights(A, m->at<uint16_t>(l.xwk), m->at<uint16_t>(l.xwv), m->H, d, st);
            }
        }
        if (xattn_supported(d, m->H)) { xattn_stream_prepare(d, 4); xattn_stream_prepare(d, 3); }
    }
    mg_stream_sync(st);    // the host tables above must outlive the copies
    const int rc = check_launch("mg_finalize");
    if (rc != MG_OK) return rc;
    m->finalized = true;
    return MG_OK;
}

int mg_attach_e1(mg_model* m, const mg_e1_model* e1) {
    if (!m) return fail(MG_E_ARG, "mg_attach_e1: null model");
    MG_ONE_CALL(m, "mg_attach_e1");
    if (!e1) { m->e1m = nullptr; m->e1_M = 0; m->step_graph.reset(); m->stream_graph.reset(); return MG_OK; }
    int tokens = 0, dm = 0, src = 0, ch = 0, fin = 0;
    e1_info(e1, &tokens, &dm, &src, &ch, &fin);
    if (!fin) return fail(MG_E_STATE, "mg_attach_e1: mg_e1_finalize has not run on the branch");
    if (dm != m->d || src != m->c.image_size || ch != m->c.num_channels)
        return fail(MG_E_SHAPE, "mg_attach_e1: the branch produces %d features from %d-channel %d px inputs, the model has d_model %d and %d-channel %d px pixel_values",
                    dm, ch, src, m->d, m->c.num_channels, m->c.image_size);
    m->e1m = e1; m->e1_M = tokens;
    m->step_graph.reset(); m->stream_graph.reset();       // captured steps hold the key-stream geometry of the previous setting
    return MG_OK;
}

int mg_workspace_bytes(const mg_model* m, int B, int L, int num_beams, int max_length, int T, int M_e1, size_t* out_bytes) {
    if (!m || !out_bytes || B < 1 || L < 1 || num_beams < 1 || max_length < 0 || T < 0 || M_e1 < 0) return fail(MG_E_ARG, "mg_workspace_bytes: bad argument");
    Ws w;
    if (M_e1 == 0) M_e1 = m->e1_M;        // attached e1 branch: sized for the calls that let the library evaluate it
    carve(m, nullptr, B, L, num_beams, max_length, T, M_e1, &w);
    *out_bytes = w.total;
    return MG_OK;
}

int mg_encode(mg_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* bbox,
              const uint8_t* attention_mask, const float* pixel_values, const float* e1, int M_e1, int B, int L, float* enc_out,
              uint8_t* enc_mask) {
    entry_drain();
    if (!m || !ws || !input_ids || !bbox || !pixel_values) return fail(MG_E_ARG, "mg_encode: null argument");
    MG_ONE_CALL(m, "mg_encode");
    if (!m->finalized) return fail(MG_E_STATE, "mg_encode: call mg_finalize first");
    if (B < 1 || L < 1) return fail(MG_E_SHAPE, "mg_encode: B and L must be >= 1");
    if ((e1 == nullptr) != (M_e1 == 0) || M_e1 < 0) return fail(MG_E_ARG, "mg_encode: e1 and M_e1 must be given together");
    mgStream_t st = (mgStream_t)stream;
    Ws w;
    const bool own_e1 = !e1 && m->e1m;            // attached OCSR branch and no precomputed tokens: evaluated here from pixel_values
    if (own_e1) M_e1 = m->e1_M;
    carve(m, (char*)ws, B, L, 1, 0, 0, M_e1, &w);
    if (w.total > ws_bytes) return fail(MG_E_WORKSPACE, "mg_encode: workspace too small (%zu < %zu)", ws_bytes, w.total);
    if (own_e1) {
        const int rc1 = e1_encode_nested(m->e1m, st, w.e1_ws, w.e1_ws_bytes, pixel_values, B, w.e1_f32, nullptr);
        if (rc1 != MG_OK) return rc1;
        e1 = w.e1_f32;
    }
    const int d = m->d, H = m->H, inner = m->inner, P = m->P;
    const int S = L + P, S_cap = round_up(S, 64), M = B * S_cap;
    mg_memset_async(w.counters, 0, 16 * sizeof(int), st);
    // patch embedding (stock:254-280) = im2col + GEMM + bias
    im2col_pack(pixel_values, w.xim, B, m->c.num_channels, m->c.image_size, m->c.patch_size, st);
    {
        GemmArgs a = gemm_args(w.xim, m->at<uint16_t>(m->patch_w), B * P, d, m->Kpatch);
        a.out_f32 = w.patch_emb; a.ldo = d; a.bias = m->at<float>(m->patch_b);
        gemm(a, EPI_F32_STORE, st);
    }
    {
        EmbedArgs e{};
        e.input_ids = input_ids; e.bbox = bbox; e.attn_mask = attention_mask; e.patch_emb = w.patch_emb;
        e.tok_emb = m->at<uint16_t>(m->tok_emb); e.x_emb = m->at<uint16_t>(m->x_emb); e.y_emb = m->at<uint16_t>(m->y_emb);
        e.B = B; e.L = L; e.P = P; e.d = d; e.n_side = m->n_side; e.M2 = m->M2; e.V = m->V; e.S_cap = S_cap;
        e.hidden = w.hidden; e.hidden_tiled = 1; e.cx = w.cx; e.cy = w.cy; e.mask = w.mask; e.xrow = w.xrow; e.xlen = w.xlen;
        e.err = w.counters + 3; e.x_row0 = M_e1; e.trim_padding = m->trim_padding ? 1 : 0; e.text_len = w.text_len;
        embed_assemble(e, w.meta, st);
    }
    // bucket indices of the three relative biases: shared by all layers and heads, computed once per batch
    bias_index(w.bidx, w.cx, w.cy, w.mask, m->at<int>(m->bk1), m->at<int>(m->bkhv), B, S, S_cap, st);
    attn_lists(w.mask, B, S, S_cap, w.att_kst, w.att_qbv, st);
    // Row tiles without any attended position are skipped by every encoder GEMM: their rows are never read as attended keys
    // (masked in the bias index / skipped stages), never produce cross K/V (xrow = -1) and their encoder output is not attended.
    // Q / K / V^T of such tiles are then never written: clear the three buffers so that a dead tile inside a live 64-key stage
    // holds finite values (its keys are masked by the -1e30 table entry, which a NaN would survive).
    const int* tiles = nullptr;
    const int* n_tiles = nullptr;
    // The list is honoured by the large-tile kernel only: it is used when EVERY encoder GEMM of this geometry runs on that kernel
    // (otherwise a GEMM on another kernel would compute dead rows from the uninitialised outputs of one that skipped them).
    int n_min = 3 * inner < d ? 3 * inner : d;
    n_min = n_min < m->dff ? n_min : m->dff;
    n_min = n_min < 2 * inner ? n_min : 2 * inner;
    const bool use_tiles = m->row_tiles && gemm_has_gelu_epilogue(M, n_min);
    m->st_row_tiles = use_tiles;
    if (use_tiles) {
        row_tile_list(w.mask, M, w.row_tiles + 1, w.row_tiles, st);
        tiles = w.row_tiles + 1; n_tiles = w.row_tiles;
        mg_memset_async(w.q_pk, 0, (size_t)M * inner * 2, st);
        mg_memset_async(w.k_pk, 0, (size_t)M * inner * 2, st);
        mg_memset_async(w.vt_pk, 0, (size_t)M * inner * 2, st);
    }
    // Encoder stack (stock:1061-1246, 644-720).  The residual stream h is fp32 in the tiled layout (ht_off); RMSNorm is
    // deferred as in the decode step: the residual projections (attention O, FFN wo) leave bf16(h * gain_next) and per-row
    // partial sums of h^2, the consuming projections (QKV, FFN wi, cross-K/V) scale their output rows by rsqrt(mean h^2 + eps).
    // Only the first norm (embedding output) and the final one (returned fp32 encoder states) are explicit launches.
    const int np4 = round_up(d / 64 > 0 ? d / 64 : 1, 4);
    mg_memset_async(w.enc_part_a, 0, (size_t)M * np4 * sizeof(float), st);
    mg_memset_async(w.enc_part_b, 0, (size_t)M * np4 * sizeof(float), st);
    const RowScale rs_a{w.enc_part_a, np4, 1.0f / (float)d, m->c.layer_norm_epsilon};     // after the FFN output
    const RowScale rs_b{w.enc_part_b, np4, 1.0f / (float)d, m->c.layer_norm_epsilon};     // after the attention output
    rmsnorm_pack_tiled(w.hidden, m->at<float>(m->enc[0].ln0), w.x_pk, nullptr, M, d, m->c.layer_norm_epsilon, st);
    for (size_t li = 0; li < m->enc.size(); ++li) {
        const EncLayer& l = m->enc[li];
        const bool last = li + 1 == m->enc.size();
        GemmArgs a = gemm_args(w.x_pk, m->at<uint16_t>(l.wqkv), M, 3 * inner, d);
        set_heads(a, H, S_cap, S_cap, w.q_pk, HF_PK_ROWS, w.k_pk, HF_PK_ROWS, w.vt_pk, HF_PK_T);
        if (li > 0) a.rs = rs_a;                   // layer 0 reads the explicitly normalised embedding
        a.row_tiles = tiles; a.n_row_tiles = n_tiles;
        gemm(a, EPI_HEADS, st);
        AttnArgs t{};
        t.Q = w.q_pk; t.K = w.k_pk; t.Vt = w.vt_pk; t.ctx = w.ctx_pk; t.B = B; t.H = H; t.Sq = S; t.Sk = S;
        t.Sq_cap = S_cap; t.Sk_cap = S_cap; t.mode = ATT_ENC; t.kmask = w.mask;
        t.tab1 = m->at<float>(m->rb_raw[0]); t.tab1_len = 32; t.tabh = m->at<float>(m->rb_raw[1]); t.tabv = m->at<float>(m->rb_raw[2]);
        t.bidx = w.bidx; t.bk1 = m->at<int>(m->bk1); t.kst = w.att_kst; t.qbv = w.att_qbv;
        attention(t, st);
        GemmArgs o = gemm_args(w.ctx_pk, m->at<uint16_t>(l.wo), M, d, inner);       // h += Wo ctx; x = bf16(h * ln1); partials -> b
        o.out_f32 = w.hidden; o.gain = m->at<float>(l.ln1); o.out_pk = w.x_pk; o.part = w.enc_part_b; o.ldo = np4;
        o.row_tiles = tiles; o.n_row_tiles = n_tiles;
        gemm(o, EPI_RESID_NORM, st);
        GemmArgs f = gemm_args(w.x_pk, m->at<uint16_t>(l.wi), M, m->dff, d);
        f.out_pk = w.y_pk; f.rs = rs_b;
        f.row_tiles = tiles; f.n_row_tiles = n_tiles;
        gemm(f, EPI_PK_RELU, st);
        GemmArgs g = gemm_args(w.y_pk, m->at<uint16_t>(l.wo2), M, d, m->dff);        // h += Wo2 y; x = bf16(h * next ln0); partials -> a
        g.out_f32 = w.hidden; g.ldo = np4;
        if (!last) { g.gain = m->at<float>(m->enc[li + 1].ln0); g.out_pk = w.x_pk; g.part = w.enc_part_a; }
        g.row_tiles = tiles; g.n_row_tiles = n_tiles;
        gemm(g, EPI_RESID_NORM, st);
    }
    rmsnorm_pack_tiled(w.hidden, m->at<float>(m->enc_ln), w.enc_pk, w.enc_f32, M, d, m->c.layer_norm_epsilon, st);
    const int* tl = m->trim_padding ? w.text_len : nullptr;
    if (enc_out) MG_LAUNCH(copy_rows_kernel, dim3(1024), dim3(256), 0, st, (const float*)w.enc_f32, enc_out, B, S, S_cap, d, L, tl);
    if (enc_mask) MG_LAUNCH(copy_bytes_rows_kernel, dim3(64), dim3(256), 0, st, (const uint8_t*)w.mask, enc_mask, B, S, S_cap, L, tl);
    // e1 tokens (OCSR vision branch, precomputed by the caller): fused with the VTL states by concatenation in front of the
    // decoder (ref: README.md:212-215) - here: packed next to them for the cross-K/V projections, never normalised or mixed
    if (e1) pack_e1(e1, B, M_e1, round_up(M_e1, 64), d, w.e1_pk, w.e1_map, w.mask, S_cap, w.xmask, st);
    m->st_B = B; m->st_L = L; m->st_S = S; m->st_Scap = S_cap; m->st_M = M_e1; m->st_ws = ws;
    return check_launch("mg_encode");
}

int mg_decoder_forward(mg_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* decoder_input_ids,
                       const uint8_t* decoder_attention_mask, int B, int T, float* logits) {
    entry_drain();
    if (!m || !ws || !decoder_input_ids || !logits) return fail(MG_E_ARG, "mg_decoder_forward: null argument");
    MG_ONE_CALL(m, "mg_decoder_forward");
    if (m->st_ws != ws || m->st_B != B) return fail(MG_E_STATE, "mg_decoder_forward: run mg_encode on this workspace/batch first");
    if (T < 1 || T > m->T_cap) return fail(MG_E_SHAPE, "mg_decoder_forward: T must be in [1, %d]", m->T_cap);
    mgStream_t st = (mgStream_t)stream;
    Ws w;
    carve(m, (char*)ws, B, m->st_L, 1, 0, T, m->st_M, &w);
    if (w.total > ws_bytes) return fail(MG_E_WORKSPACE, "mg_decoder_forward: workspace too small (%zu < %zu)", ws_bytes, w.total);
    const int d = m->d, H = m->H, inner = m->inner;
    const int T_cap = round_up(T, 64), MT = B * T_cap, S = m->st_S, S_cap = m->st_Scap, M = B * S_cap;
    const int M64 = m->st_M > 0 ? round_up(m->st_M, 64) : 0, Sx_cap = S_cap + M64;
    MG_LAUNCH(pad_dec_inputs_kernel, dim3(64), dim3(256), 0, st, decoder_input_ids, decoder_attention_mask, w.tf_ids, w.tf_mask,
              w.tf_rowmap, B, T, T_cap);
    embed_rows(w.tf_ids, m->at<uint16_t>(m->tok_emb), w.tf_hidden, MT, d, m->V, w.counters + 3, st);
    for (size_t li = 0; li < m->dec.size(); ++li) {
        const DecLayer& l = m->dec[li];
        rmsnorm_pack(w.tf_hidden, m->at<float>(l.ln0), w.tf_x, nullptr, MT, d, m->c.layer_norm_epsilon, 1.0f, st);
        GemmArgs a = gemm_args(w.tf_x, m->at<uint16_t>(l.wqkv), MT, 3 * inner, d);
        set_heads(a, H, T_cap, T_cap, w.tf_q, HF_PK_ROWS, w.tf_k, HF_PK_ROWS, w.tf_vt, HF_PK_T);
        gemm(a, EPI_HEADS, st);
        AttnArgs t{};
        t.Q = w.tf_q; t.K = w.tf_k; t.Vt = w.tf_vt; t.ctx = w.tf_ctx; t.B = B; t.H = H; t.Sq = T; t.Sk = T;
        t.Sq_cap = T_cap; t.Sk_cap = T_cap; t.mode = ATT_DEC_SELF; t.kmask = w.tf_mask;
        t.tab1 = m->at<float>(m->dec_tab); t.tab1_len = m->T_cap;
        attention(t, st);
        GemmArgs o = gemm_args(w.tf_ctx, m->at<uint16_t>(l.wo), MT, d, inner);
        o.out_f32 = w.tf_hidden; o.ldo = d;
        gemm(o, EPI_F32_RESID, st);
        // cross-attention (stock:611-640): queries from the decoder, keys/values from the encoder output
        rmsnorm_pack(w.tf_hidden, m->at<float>(l.ln1), w.tf_x, nullptr, MT, d, m->c.layer_norm_epsilon, 1.0f, st);
        GemmArgs q = gemm_args(w.tf_x, m->at<uint16_t>(l.xq), MT, inner, d);
        set_heads(q, H, T_cap, T_cap, w.tf_q, HF_PK_ROWS, nullptr, HF_NONE, nullptr, HF_NONE);
        gemm(q, EPI_HEADS, st);
        // keys / values: [e1 tokens (M64 slots) | encoder positions] side by side per image
        if (M64) {
            GemmArgs ke = gemm_args(w.e1_pk, m->at<uint16_t>(l.xkv), B * M64, 2 * inner, d);
            set_heads(ke, H, M64, Sx_cap, w.tf_xk, HF_PK_ROWS, w.tf_xvt, HF_PK_T, nullptr, HF_NONE);
            gemm(ke, EPI_HEADS, st);
        }
        GemmArgs kv = gemm_args(w.enc_pk, m->at<uint16_t>(l.xkv), M, 2 * inner, d);
        set_heads(kv, H, S_cap, Sx_cap, w.tf_xk, HF_PK_ROWS, w.tf_xvt, HF_PK_T, nullptr, HF_NONE);
        kv.heads.s_off = M64;
        gemm(kv, EPI_HEADS, st);
        AttnArgs x{};
        x.Q = w.tf_q; x.K = w.tf_xk; x.Vt = w.tf_xvt; x.ctx = w.tf_ctx; x.B = B; x.H = H; x.Sq = T; x.Sk = M64 + S;
        x.Sq_cap = T_cap; x.Sk_cap = Sx_cap; x.mode = ATT_CROSS; x.kmask = M64 ? w.xmask : w.mask;
        attention(x, st);
        GemmArgs xo = gemm_args(w.tf_ctx, m->at<uint16_t>(l.xo), MT, d, inner);
        xo.out_f32 = w.tf_hidden; xo.ldo = d;
        gemm(xo, EPI_F32_RESID, st);
        ffn_block(m, false, w.tf_hidden, w.tf_x, w.tf_y, MT, l.ln2, l.wi, l.wo2, st);
    }
    // final norm, d_model^-0.5 (tied head, stock:1554-1555), lm_head on the B*T real positions only
    rmsnorm_pack_rows(w.tf_hidden, m->at<float>(m->dec_ln), w.tf_xc, w.tf_rowmap, MT, d, m->c.layer_norm_epsilon,
                      m->tied ? 1.0f / sqrtf((float)d) : 1.0f, st);
    GemmArgs lg = gemm_args(w.tf_xc, m->at<uint16_t>(m->lm_head), B * T, m->V, d);
    lg.out_f32 = logits; lg.ldo = m->V;
    gemm(lg, EPI_F32_STORE, st);
    // out-of-range token ids (encoder or decoder side) were replaced by id 0 and counted: report them, as the
    // reference's embedding lookup would raise (SYNCHRONISES)
    int bad_ids = 0;
    mg_memcpy_async(&bad_ids, w.counters + 3, sizeof(int), st);
    mg_stream_sync(st);
    const int rc = check_launch("mg_decoder_forward");
    if (rc != MG_OK) return rc;
    if (bad_ids != 0) return fail(MG_E_INPUT, "mg_decoder_forward: %d token ids outside [0, vocab)", bad_ids);
    return MG_OK;
}

int mg_generate(mg_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* bbox,
                const uint8_t* attention_mask, const float* pixel_values, const float* e1, int M_e1, int B, int L, int num_beams, int max_length,
                int min_length, float length_penalty, int early_stopping, int64_t* out_ids, int* out_cols_host,
                float* out_scores, float* step_top2) {
    entry_drain();
    if (!m || !out_ids || !out_cols_host) return fail(MG_E_ARG, "mg_generate: null argument");
    MG_ONE_CALL(m, "mg_generate");
    if (max_length < 2 || max_length > m->T_cap) return fail(MG_E_SHAPE, "mg_generate: max_length must be in [2, %d]", m->T_cap);
    if (num_beams < 1 || num_beams > 8) return fail(MG_E_UNSUPPORTED, "mg_generate: num_beams must be in [1, 8]");
    // the decode-step projections keep all live rows of a workgroup's feature slice in registers: at most 8 row tiles
    if ((long)B * num_beams > 256)
        return fail(MG_E_UNSUPPORTED, "mg_generate: B * num_beams = %ld live sequences exceeds the supported 256; split the batch",
                    (long)B * num_beams);
    mgStream_t st = (mgStream_t)stream;
    const int K = num_beams;
    Ws w;
    if ((e1 == nullptr) != (M_e1 == 0) || M_e1 < 0) return fail(MG_E_ARG, "mg_generate: e1 and M_e1 must be given together");
    const int M_in = M_e1;
    if (!e1 && m->e1m) M_e1 = m->e1_M;            // attached OCSR branch: mg_encode evaluates it (same workspace layout as with precomputed tokens)
    carve(m, (char*)ws, B, L, K, max_length, 0, M_e1, &w);
    if (w.total > ws_bytes) return fail(MG_E_WORKSPACE, "mg_generate: workspace too small (%zu < %zu)", ws_bytes, w.total);
    if (m->phase_on) mg_event_record(m->phase_ev[0], st);
    int rc = mg_encode(m, stream, ws, ws_bytes, input_ids, bbox, attention_mask, pixel_values, e1, M_in, B, L, nullptr, nullptr);
    if (rc != MG_OK) return rc;
    const int d = m->d, H = m->H, inner = m->inner, S_cap = m->st_Scap, M = B * S_cap;
    const int R = B * K, T_cap = m->T_cap;
    const size_t nl = m->dec.size();
    const int M64 = M_e1 > 0 ? round_up(M_e1, 64) : 0, Sx_cap = S_cap + M64;
    const size_t xkv_stride = (size_t)B * H * Sx_cap * 64, skv_stride = (size_t)R * H * T_cap * 64;
    // cross-attention K/V of every decoder layer, once per image (stock:524-538), compacted to attended positions; the
    // e1 tokens (if any) occupy rows [0, M_e1) of an image's stream, the attended encoder positions follow (xrow carries
    // the offset) - cross-attention has no positional term, so the order of the keys is immaterial
    const bool absorbed = use_absorb(m, K, R);
    if (absorbed) {
        // weight-absorbed form (k_xattn.hip): the decoder layers stream the attended states themselves - one compaction instead of
        // 2 x N_dec projections
        if (M64) enc_rows(w.e1_pk, w.e1_map, w.encx, B, M64, Sx_cap, d, st);
        enc_rows(w.enc_pk, w.xrow, w.encx, B, S_cap, Sx_cap, d, st);
        enc_pad_rows(w.encx, w.xlen, B, Sx_cap, d, st);
    }
    for (size_t li = 0; li < nl && !absorbed; ++li) {
        if (M64) {
            GemmArgs ke = gemm_args(w.e1_pk, m->at<uint16_t>(m->dec[li].xkv), B * M64, 2 * inner, d);
            set_heads(ke, H, M64, Sx_cap, w.xk + li * xkv_stride, HF_NATURAL, w.xv + li * xkv_stride, HF_NATURAL, nullptr, HF_NONE);
            ke.heads.row_map = w.e1_map;
            gemm(ke, EPI_HEADS, st);
        }
        GemmArgs kv = gemm_args(w.enc_pk, m->at<uint16_t>(m->dec[li].xkv), M, 2 * inner, d);
        set_heads(kv, H, S_cap, Sx_cap, w.xk + li * xkv_stride, HF_NATURAL, w.xv + li * xkv_stride, HF_NATURAL, nullptr, HF_NONE);
        kv.heads.row_map = w.xrow;
        if (m->st_row_tiles) { kv.row_tiles = w.row_tiles + 1; kv.n_row_tiles = w.row_tiles; }      // left by mg_encode
        gemm(kv, EPI_HEADS, st);
    }
    if (m->phase_on) mg_event_record(m->phase_ev[1], st);
    int* counters = w.counters;
    const int64_t start = m->c.decoder_start_token_id, pad = m->c.pad_token_id;
#ifndef MG_EMU
    // The legacy null stream cannot be captured: the decode phase then runs on a stream the model owns, ordered after
    // the caller's stream by an event (the call ends with a host synchronisation of that stream, which orders it
    // before anything the caller enqueues later).
    if (m->use_graph == 1 && st == nullptr) {
        if (!m->own_stream && hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess) m->own_stream = nullptr;
        if (!m->fork_ev && hipEventCreateWithFlags(&m->fork_ev, hipEventDisableTiming) != hipSuccess) m->fork_ev = nullptr;
        if (m->own_stream && m->fork_ev && hipEventRecord(m->fork_ev, st) == hipSuccess &&
            hipStreamWaitEvent(m->own_stream, m->fork_ev, 0) == hipSuccess)
            st = m->own_stream;
    }
#endif
    if (K == 1) {
        MG_LAUNCH(fill_ids_kernel, dim3(R), dim3(64), 0, st, w.next_ids, out_ids, w.unfinished, counters, R, max_length, start, pad);
    } else {
        beam_init(w.beam_state, B, K, max_length, (int)pad, m->c.eos_token_id, (int)start, w.next_ids, w.anc, T_cap, counters, st);
        m->beam_div_host.resize((size_t)max_length + 1);
        for (int c = 0; c <= max_length; ++c) m->beam_div_host[c] = beam_length_divisor(c, length_penalty);
        mg_memcpy_async(w.beam_div, m->beam_div_host.data(), m->beam_div_host.size() * sizeof(float), st);
    }
    int steps_done = 0;
    int host_flag[4] = {0, 0, 0, 0};
    // greedy with EOS enabled: finished rows emit pad whatever they compute, their attention launches skip them
    // (not under the parity instrumentation, which compares every row's logits at every captured step)
    const int* live = (K == 1 && min_length < max_length && !m->dbg_logits && !m->dbg_forced) ? w.unfinished : nullptr;
    DecodeCtx dc{};
    dc.xk = w.xk; dc.xv = w.xv; dc.xkv_stride = xkv_stride; dc.Sx_cap = Sx_cap; dc.xlen = w.xlen;
    dc.encx = w.encx; dc.qx = w.qx; dc.xpart = w.xpart; dc.xml = w.xml;
    dc.sk = w.sk; dc.sv = w.sv; dc.skv_stride = skv_stride;
    dc.dq = w.dq; dc.dx_pk = w.dx_pk; dc.dy_pk = w.dy_pk; dc.xa = w.xa; dc.xb = w.xb;
    dc.dh = w.dh; dc.logits = w.logits; dc.rs_part = w.rs_part; dc.rs_part1 = w.rs_part1; dc.rs_part2 = w.rs_part2;
    dc.kpart = w.slabs; dc.tickets = w.tickets;
    mg_memset_async(w.tickets, 0, 512 * sizeof(int), st);
    dc.next_ids = w.next_ids; dc.unfinished = w.unfinished; dc.anc = w.anc; dc.beam_idx = w.beam_idx; dc.beam_div = w.beam_div;
    dc.beam_state = w.beam_state; dc.counters = counters;
    dc.B = B; dc.K = K; dc.R = R; dc.max_length = max_length; dc.min_length = min_length; dc.early_stopping = early_stopping;
    dc.length_penalty = length_penalty; dc.out_ids = out_ids; dc.step_top2 = step_top2; dc.live = live;
    // greedy batch calls run the fused tail (lm_head top-2 partials -> selection + next embedding in one launch); the parity
    // instrumentation needs the full logits / overrides the fed token, and d_model > 2048 would change the norm's summation order
    const bool fused_tail = K == 1 && m->fused_tail && !m->dbg_logits && !m->dbg_forced && d <= 2048;
    if (fused_tail) {
        dc.ptop = w.ptop; dc.stopv = w.stopv;
        embed_norm_rows(w.next_ids, m->at<uint16_t>(m->tok_emb), w.dh, m->at<float>(m->dec[0].ln0), w.dx_pk, w.xa, d + inner, 0, R, d, m->V,
                        counters + 3, m->c.layer_norm_epsilon, st);      // the start token; later steps: greedy_select_fused
    }
    auto decode_step = [&](int t, const int* tdev, bool time_cross) { ::decode_step(m, dc, t, tdev, time_cross, st); };
    bool graphed = false;
    const bool instrumented = m->dbg_logits || m->dbg_forced;      // by-value eager launches of the same kernels
#ifndef MG_EMU
    if (m->use_graph == 1 && !instrumented) {
        const StepGraph::Key key{ws, out_ids, step_top2, (const void*)st, B, L, K, max_length, min_length, early_stopping, M_e1, length_penalty, nullptr};
        StepGraph& sg = m->step_graph;
        if (!(sg.valid && sg.key == key)) {
            std::lock_guard<std::mutex> capture_lock(mg_capture_mutex());
            sg.reset();
            hipGraph_t graph = nullptr;
            hipError_t e1 = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal), e2 = hipSuccess, e3 = hipSuccess;
            if (e1 == hipSuccess) {
                decode_step(0, counters + 2, false);
                e2 = hipStreamEndCapture(st, &graph);
                if (e2 == hipSuccess && graph) {
                    e3 = hipGraphInstantiate(&sg.exec, graph, nullptr, nullptr, 0);
                    if (e3 == hipSuccess) { sg.key = key; sg.valid = true; }
                }
                if (graph) (void)hipGraphDestroy(graph);
            }
            if (!sg.valid && getenv("MG_DEBUG"))
                fprintf(stderr, "mg_generate: decode-step capture failed (begin %s, end %s, instantiate %s); launching eagerly\n",
                        hipGetErrorName(e1), hipGetErrorName(e2), hipGetErrorName(e3));
            (void)hipGetLastError();
        }
        graphed = sg.valid;
    }
#endif
    m->graph_active = graphed;
    // The host enqueues steps far faster than the GPU runs them; a full runtime queue makes the launching thread SPIN inside the launch
    // call (one busy core per execution context).  Pace it instead: every 8 steps an event is recorded, and before enqueueing further the
    // thread sleeps (polling an event: mg_event_sync) until the steps of 16 .. 24 steps ago have run - the GPU always has >= 16 steps queued.
    constexpr int PACE = 8, PACE_RING = 3;
    while ((int)m->pace_ev.size() < PACE_RING) { mgEvent_t e; if (mg_event_create_notiming(&e) != 0) break; m->pace_ev.push_back(e); }
    const bool paced = (int)m->pace_ev.size() == PACE_RING && m->pace;
    for (int t = 0; t + 1 < max_length; ++t) {
        if (paced && (t % PACE) == 0) {
            const int k = t / PACE;
            if (k >= PACE_RING) mg_event_sync(m->pace_ev[k % PACE_RING]);      // recorded PACE_RING * PACE steps ago
            mg_event_record(m->pace_ev[k % PACE_RING], st);
        }
        const bool timed_step = m->prof_every > 0 && (t % m->prof_every) == 0;
#ifndef MG_EMU
        if (graphed && !timed_step) {
            if (hipGraphLaunch(m->step_graph.exec, st) != hipSuccess) return fail(MG_E_HIP, "mg_generate: hipGraphLaunch failed");
        } else
#endif
        {
            // use_graph == 2: the device-counter form launched eagerly (what the graph replays; testable without HIP graphs).
            // A timed step of a graphed call launches that same device-counter form, so the bracketed launches are the
            // kernels the graph replays (the step counter lives on the device and advances identically).
            const bool dev_form = !instrumented && (m->use_graph == 2 || graphed);
            decode_step(t, dev_form ? counters + 2 : nullptr, timed_step);
        }
        steps_done = t + 1;
        // termination is checked every 8 steps (and at the end): overrunning only appends pad columns, which are
        // trimmed with the device-recorded `done_step`.  With min_length >= max_length EOS is suppressed at every
        // position, no row can finish early and the host never looks.
        if (min_length < max_length && ((t & 7) == 7 || t + 2 >= max_length)) {
            mg_memcpy_async(host_flag, counters, sizeof host_flag, st);
            mg_stream_sync(st);
            if (host_flag[0] == 0) break;
        }
    }
    if (K > 1) beam_finalize(w.beam_state, B, K, max_length, out_ids, counters + 4, out_scores, st);
    if (m->phase_on) mg_event_record(m->phase_ev[2], st);
    std::vector<int> xlen_host;
    if (m->prof_used) { xlen_host.resize(B); mg_memcpy_async(xlen_host.data(), w.xlen, (size_t)B * sizeof(int), st); }
    mg_memcpy_async(host_flag, counters, sizeof host_flag, st);
    int beam_cols = 0;
    if (K > 1) mg_memcpy_async(&beam_cols, counters + 4, sizeof(int), st);
    mg_stream_sync(st);
    rc = check_launch("mg_generate");
    if (rc != MG_OK) return rc;
    if (m->prof_used) {
        double keys = 0.0;
        for (int b = 0; b < B; ++b) keys += xlen_host[b];
        for (size_t i = 0; i + 2 < m->prof_used; i += 3) {
            m->prof_ms += mg_event_elapsed_ms(m->prof_ev[i], m->prof_ev[i + 1]);
            m->prof_empty_ms += mg_event_elapsed_ms(m->prof_ev[i + 1], m->prof_ev[i + 2]);
            m->prof_n += 1;
            m->prof_keys += keys;
        }
        m->prof_used = 0;
    }
    if (m->phase_on) {
        m->phase_enc_ms += mg_event_elapsed_ms(m->phase_ev[0], m->phase_ev[1]);
        m->phase_dec_ms += mg_event_elapsed_ms(m->phase_ev[1], m->phase_ev[2]);
        m->phase_n += 1;
    }
    if (host_flag[3] != 0) return fail(MG_E_INPUT, "mg_generate: %d token ids outside [0, vocab)", host_flag[3]);
    if (K == 1) *out_cols_host = 1 + (host_flag[1] >= 0 ? host_flag[1] + 1 : steps_done);
    else *out_cols_host = beam_cols;
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Continuous greedy decoding of N images (SURVEY.md section 7 "early-exit compaction of finished sequences", section 8e
// stragglers).  What the reference does one image at a time - generate(max_length=512) until EOS (utils_evaluation.py:269-285) -
// on `slots` decode rows that work through the queue of images: a row that ends frees its slot for the next image, so no step
// is spent on finished rows and the batch never waits for its longest member.  The encoder + cross-K/V projection of the next
// `chunk` images run AHEAD on a second stream (low priority or CU-masked), under the launch-bound decode steps of the current
// ones; their K/V land in a pool of pool_chunks x chunk entries that the decode rows read through the slot table.  All slot
// bookkeeping is on the device (k_decode.hip: selection + slot_refill), the step is one replayed HIP graph; the host only
// feeds the encoder stream and reads the counters back a few steps late.  Every image's ids equal what mg_generate returns for
// it: rows of the decode kernels are independent of each other and of the slot they sit in (tests/test_stream.py).
int mg_stream_workspace_bytes(const mg_model* m, int chunk, int L, int slots, int pool_chunks, size_t* out_bytes) {
    if (!m || !out_bytes || chunk < 1 || L < 1 || slots < 1 || pool_chunks < 2) return fail(MG_E_ARG, "mg_stream_workspace_bytes: bad argument");
    StreamWs w;
    carve_stream(m, nullptr, chunk, L, slots, pool_chunks, &w);
    *out_bytes = w.total;
    return MG_OK;
}

// mode 0: encoder on the caller's stream (serial: for A/B runs and the emulator); 1 (default): own stream at the lowest
// priority; 2: own stream restricted to the compute units of cu_mask (nwords x 32 bits).  Takes effect at the next call.
int mg_stream_encoder_mode(mg_model* m, int mode, const uint32_t* cu_mask, int nwords) {
    if (!m || mode < 0 || mode > 2 || (mode == 2 && (!cu_mask || nwords < 1))) return fail(MG_E_ARG, "mg_stream_encoder_mode: bad argument");
    MG_ONE_CALL(m, "mg_stream_encoder_mode");          // (a call on this context may be using the stream that is destroyed below)
    if (m->enc_stream_ready && m->enc_stream) { mg_stream_sync(m->enc_stream); mg_stream_destroy(m->enc_stream); }
    m->enc_stream = nullptr; m->enc_stream_ready = false;
    m->enc_mode = mode;
    m->enc_mask.assign(cu_mask, cu_mask + (mode == 2 ? nwords : 0));
    return MG_OK;
}

}   // extern "C"

// Continuous decoding of a queue of images: greedy (K = 1: `slots` decode rows) or beam search (K > 1: `slots` IMAGE slots of K rows).
static int generate_stream_impl(mg_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* bbox,
                                const uint8_t* attention_mask, const float* pixel_values, int N, int L, int chunk, int slots, int pool_chunks,
                                int K, int max_length, int min_length, float length_penalty, int early_stopping, int64_t* out_ids,
                                int32_t* out_len, float* out_scores, long* steps_host, const char* who) {
    entry_drain();
    if (!m || !ws || !input_ids || !bbox || !pixel_values || !out_ids || !out_len) return fail(MG_E_ARG, "%s: null argument", who);
    MG_ONE_CALL(m, "mg_generate_stream");
    if (!m->finalized) return fail(MG_E_STATE, "%s: call mg_finalize first", who);
    if (N < 1 || L < 1 || chunk < 1 || pool_chunks < 2) return fail(MG_E_SHAPE, "%s: N, L, chunk must be >= 1, pool_chunks >= 2", who);
    if (K < 1 || K > 8) return fail(MG_E_UNSUPPORTED, "%s: num_beams must be in [1, 8]", who);
    if (slots < 1 || (long)slots * K > 256) return fail(MG_E_UNSUPPORTED, "%s: slots * num_beams must be in [1, 256]", who);
    if (slots > pool_chunks * chunk) return fail(MG_E_SHAPE, "%s: slots (%d) exceed the %d pool entries", who, slots, pool_chunks * chunk);
    if (max_length < 2 || max_length > m->T_cap) return fail(MG_E_SHAPE, "%s: max_length must be in [2, %d]", who, m->T_cap);
    if (m->dbg_logits || m->dbg_forced) return fail(MG_E_STATE, "%s: the decode-capture instrumentation is for mg_generate", who);
    const int R = slots * K;                         // decode rows
    StreamWs w;
    carve_stream(m, (char*)ws, chunk, L, slots, pool_chunks, &w, K, max_length);
    if (w.total > ws_bytes) return fail(MG_E_WORKSPACE, "%s: workspace too small (%zu < %zu)", who, ws_bytes, w.total);
    mgStream_t st = (mgStream_t)stream;
    const int d = m->d, H = m->H, inner = m->inner, P = m->P;
    const int S_cap = round_up(L + P, 64), M64 = round_up(m->e1_M, 64), Sx_cap = S_cap + M64;
    const size_t nl = m->dec.size();
    const int n_chunks = (N + chunk - 1) / chunk;
    const int entries = pool_chunks * chunk;
    const size_t img_in = (size_t)m->c.num_channels * m->c.image_size * m->c.image_size;
    // streams and events
    if (m->enc_mode != 0 && !m->enc_stream_ready) {
        if (mg_stream_create(&m->enc_stream, m->enc_mode == 1, m->enc_mode == 2 ? m->enc_mask.data() : nullptr, (int)m->enc_mask.size()) != 0)
            return fail(MG_E_HIP, "%s: could not create the encoder stream", who);
        m->enc_stream_ready = true;
    }
#ifndef MG_EMU
    if (st == nullptr) {       // the legacy null stream synchronises with every other stream and cannot be captured: own stream
        if (!m->own_stream && hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess) m->own_stream = nullptr;
        if (!m->fork_ev && hipEventCreateWithFlags(&m->fork_ev, hipEventDisableTiming) != hipSuccess) m->fork_ev = nullptr;
        if (m->own_stream && m->fork_ev && hipEventRecord(m->fork_ev, st) == hipSuccess && hipStreamWaitEvent(m->own_stream, m->fork_ev, 0) == hipSuccess)
            st = m->own_stream;
    }
#endif
    mgStream_t es = m->enc_mode == 0 ? st : m->enc_stream;
    while ((int)m->chunk_ev.size() < 2 * pool_chunks + 2) { mgEvent_t e; if (mg_event_create(&e) != 0) return fail(MG_E_HIP, "event"); m->chunk_ev.push_back(e); }
    constexpr int RB = 4, GROUP = 4;          // read-back ring depth, steps per host iteration
    while ((int)m->rb_ev.size() < RB) { mgEvent_t e; if (mg_event_create_notiming(&e) != 0) return fail(MG_E_HIP, "event"); m->rb_ev.push_back(e); }
    if (!m->start_ev && mg_event_create_notiming(&m->start_ev) != 0) return fail(MG_E_HIP, "event");
    if (!m->stream_host && !(m->stream_host = (int*)mg_host_alloc(RB * 16 * sizeof(int)))) return fail(MG_E_HIP, "pinned host buffer");
    // slot table, outputs
    const int64_t start = m->c.decoder_start_token_id, pad = m->c.pad_token_id;
    MG_LAUNCH(stream_init_kernel, dim3(64), dim3(256), 0, st, out_ids, out_len, N, max_length, start, pad, w.unfinished, w.pos, w.img, w.pool,
              w.next_ids, R, w.ctr, w.err);
    mg_memset_async(w.bpool, 0, (size_t)round_up(slots, 32) * sizeof(int), st);
    mg_memset_async(w.assign, 0xFF, (size_t)round_up(slots, 32) * sizeof(int), st);
    mg_memset_async(w.xlen_pool, 0, (size_t)entries * sizeof(int), st);           // (idle slots' cross-attention is skipped; belt and braces)
    if (K > 1) {
        m->beam_div_host.resize((size_t)max_length + 1);
        for (int c = 0; c <= max_length; ++c) m->beam_div_host[c] = beam_length_divisor(c, length_penalty);
        mg_memcpy_async(w.beam_div, m->beam_div_host.data(), m->beam_div_host.size() * sizeof(float), st);
        mg_stream_sync(st);        // (the host vector may be resized by the context's next call)
    }
    mg_event_record(m->start_ev, st);
    if (es != st) mg_stream_wait_event(es, m->start_ev);      // inputs / workspace are ordered behind the caller's earlier work
    DecodeCtx dc{};
    dc.xk = w.xk; dc.xv = w.xv; dc.xkv_stride = w.pool_stride; dc.Sx_cap = Sx_cap; dc.xlen = w.xlen_pool;
    dc.encx = w.encx; dc.qx = w.qx; dc.xpart = w.xpart; dc.xml = w.xml;
    dc.sk = w.sk; dc.sv = w.sv; dc.skv_stride = (size_t)R * H * m->T_cap * 64;
    dc.dq = w.dq; dc.dx_pk = w.dx_pk; dc.dy_pk = w.dy_pk; dc.xa = w.xa; dc.xb = w.xb;
    dc.dh = w.dh; dc.logits = w.logits; dc.rs_part = w.rs_part; dc.rs_part1 = w.rs_part1; dc.rs_part2 = w.rs_part2;
    dc.kpart = w.kpart; dc.tickets = w.tickets;
    mg_memset_async(w.tickets, 0, 512 * sizeof(int), st);
    dc.next_ids = w.next_ids; dc.unfinished = w.unfinished; dc.counters = w.ctr;
    dc.B = slots; dc.K = K; dc.R = R; dc.max_length = max_length; dc.min_length = min_length; dc.length_penalty = length_penalty;
    dc.early_stopping = early_stopping;
    dc.out_ids = out_ids; dc.live = w.unfinished;
    dc.anc = w.anc; dc.beam_idx = w.beam_idx; dc.beam_div = w.beam_div; dc.beam_state = w.beam_state;
    dc.bpool = w.bpool; dc.assign = w.assign; dc.out_len = out_len; dc.out_scores = out_scores;
    dc.slots = SlotTable{w.pos, w.img, w.pool, w.ctr, out_len, entries, (int)start};
    // the step as a graph (every step-dependent value lives in the slot table)
    bool graphed = false;
#ifndef MG_EMU
    if (m->use_graph == 1) {
        const StepGraph::Key key{ws, out_ids, out_len, (const void*)st, slots * 16 + K, L, chunk, max_length, min_length, N * 2 + (early_stopping ? 1 : 0),
                                 pool_chunks, length_penalty, (const void*)out_scores};     // (B = slots and beams, K = chunk, early_stopping = N and the flag, M_e1 = pool_chunks)
        StepGraph& sg = m->stream_graph;
        if (!(sg.valid && sg.key == key)) {
            std::lock_guard<std::mutex> capture_lock(mg_capture_mutex());
            sg.reset();
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                decode_step(m, dc, 0, nullptr, false, st);
                if (hipStreamEndCapture(st, &graph) == hipSuccess && graph && hipGraphInstantiate(&sg.exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    sg.key = key; sg.valid = true;
                }
                if (graph) (void)hipGraphDestroy(graph);
            }
            (void)hipGetLastError();
        }
        graphed = sg.valid;
    }
#endif
    m->graph_active = graphed;
    int submitted = 0, announced = 0;           // chunks handed to the encoder stream / made visible to the slots
    int oldest_host = 0, done_host = 0, live_host = 0, head_host = 0;
    long steps = 0, rb_issued = 0, rb_seen = 0;
    m->stream_enc_ms = 0.0;
    auto submit_chunk = [&]() -> int {
        const int c0 = submitted * chunk, n = (N - c0) < chunk ? (N - c0) : chunk;
        const int entry0 = (submitted % pool_chunks) * chunk;
        mgEvent_t e0 = m->chunk_ev[2 * (submitted % pool_chunks)], e1 = m->chunk_ev[2 * (submitted % pool_chunks) + 1];
        mg_event_record(e0, es);
        NestedEntry nested;                    // keeps the pending runtime error / first failing site of this call's earlier launches
        int rc = mg_encode(m, es, ws, ws_bytes, input_ids + (size_t)c0 * L, bbox + (size_t)c0 * L * 4, attention_mask ? attention_mask + (size_t)c0 * L : nullptr,
                           pixel_values + (size_t)c0 * img_in, nullptr, 0, n, L, nullptr, nullptr);
        if (rc != MG_OK) return rc;
        Ws we;
        carve(m, (char*)ws, n, L, 1, 0, 0, m->e1_M, &we);         // the chunk's own carving (a short last chunk uses less of the region)
        const size_t ent_off = (size_t)entry0 * H * Sx_cap * 64;
        const bool absorbed = w.encx != nullptr;
        if (absorbed) {        // the chunk's attended states into its pool entries (as mg_generate)
            uint16_t* ex = w.encx + (size_t)entry0 * Sx_cap * d;
            if (M64) enc_rows(we.e1_pk, we.e1_map, ex, n, M64, Sx_cap, d, es);
            enc_rows(we.enc_pk, we.xrow, ex, n, S_cap, Sx_cap, d, es);
            enc_pad_rows(ex, we.xlen, n, Sx_cap, d, es);
        }
        for (size_t li = 0; li < nl && !absorbed; ++li) {
            if (M64) {          // the e1 tokens of the attached OCSR branch: rows [0, e1_M) of every image's key stream (as mg_generate)
                GemmArgs ke = gemm_args(we.e1_pk, m->at<uint16_t>(m->dec[li].xkv), n * M64, 2 * inner, d);
                set_heads(ke, H, M64, Sx_cap, w.xk + li * w.pool_stride + ent_off, HF_NATURAL, w.xv + li * w.pool_stride + ent_off, HF_NATURAL, nullptr, HF_NONE);
                ke.heads.row_map = we.e1_map;
                gemm(ke, EPI_HEADS, es);
            }
            GemmArgs kv = gemm_args(we.enc_pk, m->at<uint16_t>(m->dec[li].xkv), n * S_cap, 2 * inner, d);
            set_heads(kv, H, S_cap, Sx_cap, w.xk + li * w.pool_stride + ent_off, HF_NATURAL, w.xv + li * w.pool_stride + ent_off, HF_NATURAL, nullptr, HF_NONE);
            kv.heads.row_map = we.xrow;
            if (m->st_row_tiles) { kv.row_tiles = we.row_tiles + 1; kv.n_row_tiles = we.row_tiles; }
            gemm(kv, EPI_HEADS, es);
        }
        MG_LAUNCH(stream_chunk_done_kernel, dim3(1), dim3(64), 0, es, (const int*)we.xlen, w.xlen_pool, entry0, n, (const int*)we.counters, w.err);
        mg_event_record(e1, es);
        ++submitted;
        return MG_OK;
    };
    auto announce = [&](bool wait) {
        mgEvent_t e1 = m->chunk_ev[2 * (announced % pool_chunks) + 1];
        if (es != st) {
            if (!wait && !mg_event_done(e1)) return false;
            mg_stream_wait_event(st, e1);
        }
        const int c0 = announced * chunk, n = (N - c0) < chunk ? (N - c0) : chunk;
        MG_LAUNCH(stream_ready_kernel, dim3(1), dim3(64), 0, st, n, w.ctr);
        ++announced;
        return true;
    };
    int rc = MG_OK;
    // error returns from the loop: nothing enqueued by this call may still be writing ws / out_ids / out_len when it returns
    auto quiesce = [&]() { if (es != st) mg_stream_sync(es); mg_stream_sync(st); };
    while (done_host < N) {
        // feed the encoder stream: chunk c overwrites the pool entries of chunk c - pool_chunks, whose images must all have
        // finished (oldest live image known to the host, a few steps late: conservative)
        while (submitted < n_chunks && (submitted < pool_chunks || oldest_host >= (submitted - pool_chunks + 1) * chunk) &&
               (es != st || submitted == announced)) {
            if ((rc = submit_chunk()) != MG_OK) { quiesce(); return rc; }
            if (es == st) break;              // serial mode: one chunk, then decode until the slots run dry
        }
        // hand finished chunks to the slots; when no slot is live and the queue is empty the decode stream has to wait for one
        const bool starving = live_host == 0 && head_host >= announced * chunk;     // (late view; still true now: nothing was announced since)
        while (announced < submitted && announce(starving && announced * chunk <= head_host)) {}
        if (announced == 0) { announce(true); }
        for (int g = 0; g < GROUP; ++g) {
            const bool timed_step = m->prof_every > 0 && (steps % m->prof_every) == 0;
#ifndef MG_EMU
            if (graphed && !timed_step) {
                if (hipGraphLaunch(m->stream_graph.exec, st) != hipSuccess) { quiesce(); return fail(MG_E_HIP, "%s: hipGraphLaunch failed", who); }
            } else
#endif
                decode_step(m, dc, 0, nullptr, timed_step, st);
            ++steps;
        }
        // counters back to the host, asynchronously; look at the oldest outstanding copy only when the ring is full
        int* slot = m->stream_host + (rb_issued % RB) * 16;
        mg_memcpy_async(slot, w.ctr, 16 * sizeof(int), st);
        mg_event_record(m->rb_ev[rb_issued % RB], st);
        ++rb_issued;
        while (rb_seen < rb_issued && (rb_issued - rb_seen >= RB - 1 || mg_event_done(m->rb_ev[rb_seen % RB]))) {
            mg_event_sync(m->rb_ev[rb_seen % RB]);
            const int* h = m->stream_host + (rb_seen % RB) * 16;
            live_host = h[0]; done_host = h[1]; head_host = h[4]; oldest_host = h[7];
            ++rb_seen;
        }
        if (steps > (long)N * max_length + 64L * n_chunks + 1024) { quiesce(); return fail(MG_E_HIP, "%s: no progress (%d of %d images after %ld steps)", who, done_host, N, steps); }
    }
    int err2[2] = {0, 0};
    int& err_host = err2[0];
    if (es != st) mg_stream_sync(es);
    mg_memcpy_async(err2, w.err, 2 * sizeof(int), st);
    mg_stream_sync(st);
    rc = check_launch(who);
    if (rc != MG_OK) return rc;
    for (int c = 0; c < (n_chunks < pool_chunks ? n_chunks : pool_chunks); ++c)
        m->stream_enc_ms += mg_event_elapsed_ms(m->chunk_ev[2 * c], m->chunk_ev[2 * c + 1]);     // the last pool_chunks chunks (statistics)
    if (m->prof_used) {      // cross-attention brackets of the timed steps: keys streamed = live slots' pool entries at that step (approximated by the mean)
        for (size_t i = 0; i + 2 < m->prof_used; i += 3) {
            m->prof_ms += mg_event_elapsed_ms(m->prof_ev[i], m->prof_ev[i + 1]);
            m->prof_empty_ms += mg_event_elapsed_ms(m->prof_ev[i + 1], m->prof_ev[i + 2]);
            m->prof_n += 1;
            m->prof_keys += (double)err2[1] / N * (N < slots ? N : slots);
        }
        m->prof_used = 0;
    }
    m->stream_steps = steps;
    if (steps_host) *steps_host = steps;
    if (err_host != 0) return fail(MG_E_INPUT, "%s: %d token ids outside [0, vocab)", who, err_host);
    return MG_OK;
}

extern "C" {

int mg_generate_stream(mg_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* bbox,
                       const uint8_t* attention_mask, const float* pixel_values, int N, int L, int chunk, int slots, int pool_chunks,
                       int max_length, int min_length, int64_t* out_ids, int32_t* out_len, long* steps_host) {
    return generate_stream_impl(m, stream, ws, ws_bytes, input_ids, bbox, attention_mask, pixel_values, N, L, chunk, slots, pool_chunks, 1,
                                max_length, min_length, 1.0f, 0, out_ids, out_len, nullptr, steps_host, "mg_generate_stream");
}
int mg_stream_beam_workspace_bytes(const mg_model* m, int chunk, int L, int slots, int pool_chunks, int num_beams, int max_length, size_t* out_bytes) {
    if (!m || !out_bytes || chunk < 1 || L < 1 || slots < 1 || pool_chunks < 2 || num_beams < 1 || num_beams > 8 || max_length < 2)
        return fail(MG_E_ARG, "mg_stream_beam_workspace_bytes: bad argument");
    StreamWs w;
    carve_stream(m, nullptr, chunk, L, slots, pool_chunks, &w, num_beams, max_length);
    *out_bytes = w.total;
    return MG_OK;
}
int mg_generate_stream_beam(mg_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* bbox,
                            const uint8_t* attention_mask, const float* pixel_values, int N, int L, int chunk, int slots, int pool_chunks,
                            int num_beams, int max_length, int min_length, float length_penalty, int early_stopping, int64_t* out_ids,
                            int32_t* out_len, float* out_scores, long* steps_host) {
    return generate_stream_impl(m, stream, ws, ws_bytes, input_ids, bbox, attention_mask, pixel_values, N, L, chunk, slots, pool_chunks,
                                num_beams, max_length, min_length, length_penalty, early_stopping, out_ids, out_len, out_scores, steps_host, "mg_generate_stream_beam");
}

// Live timing of the dominant decode kernel (single-query cross-attention over the image K/V stream): when enabled,
// every `every`-th decode step brackets each layer's launch with HIP events on the caller's stream.
int mg_profile_cross_attention(mg_model* m, int every, int max_samples) {
    if (!m) return fail(MG_E_ARG, "mg_profile_cross_attention: null model");
    for (mgEvent_t e : m->prof_ev) mg_event_destroy(e);
    m->prof_ev.clear();
    m->prof_every = every;
    m->prof_used = 0; m->prof_ms = 0.0; m->prof_empty_ms = 0.0; m->prof_n = 0; m->prof_keys = 0.0;
    for (int i = 0; every > 0 && i < 3 * max_samples; ++i) {
        mgEvent_t e;
        if (mg_event_create(&e) != 0) return fail(MG_E_HIP, "hipEventCreate failed");
        m->prof_ev.push_back(e);
    }
    return MG_OK;
}
// Parity-test instrumentation of mg_generate (see mgrapher.h); cleared by passing null pointers.
int mg_debug_decode_capture(mg_model* m, float* logits_capture, int capture_steps, const int64_t* forced_ids) {
    if (!m) return fail(MG_E_ARG, "mg_debug_decode_capture: null model");
    m->dbg_logits = logits_capture;
    m->dbg_steps = logits_capture ? capture_steps : 0;
    m->dbg_forced = forced_ids;
    return MG_OK;
}
// Phase timing of mg_generate with three HIP events per call: [encoder + cross-K/V precompute | decode loop].
int mg_profile_phases(mg_model* m, int enable) {
    if (!m) return fail(MG_E_ARG, "mg_profile_phases: null model");
    if (enable && !m->phase_ev[0]) {
        for (int i = 0; i < 3; ++i)
            if (mg_event_create(&m->phase_ev[i]) != 0) return fail(MG_E_HIP, "hipEventCreate failed");
    }
    m->phase_on = enable != 0;
    m->phase_n = 0; m->phase_enc_ms = 0.0; m->phase_dec_ms = 0.0;
    return MG_OK;
}
int mg_profile_phases_read(mg_model* m, long* calls, double* enc_ms, double* dec_ms) {
    if (!m) return fail(MG_E_ARG, "mg_profile_phases_read: null model");
    if (calls) *calls = m->phase_n;
    if (enc_ms) *enc_ms = m->phase_enc_ms;
    if (dec_ms) *dec_ms = m->phase_dec_ms;
    return MG_OK;
}
// 0 (default): a batch's trailing text padding is part of the sequence, as stock HF computes a padded batch; 1: per-image semantics -
// every image's patches follow ITS last attended text token, the result for an image does not depend on how far its batch was padded
// and equals what the reference's batch-size-1 loop computes for it (utils_evaluation.py:140).  Returns the previous setting.
int mg_set_padding_semantics(mg_model* m, int per_image) {
    if (!m) return fail(MG_E_ARG, "mg_set_padding_semantics: null model");
    const int prev = m->trim_padding ? 1 : 0;
    m->trim_padding = per_image != 0;
    return prev;
}
int mg_set_decode_graph(mg_model* m, int enable) {
    if (!m) return fail(MG_E_ARG, "mg_set_decode_graph: null model");
    const int prev = m->use_graph;
    m->use_graph = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
    if (m->use_graph != 1) { m->step_graph.reset(); m->stream_graph.reset(); }      // (both captured steps: batch form and the greedy / beam queues)
    return prev;
}
int mg_decode_graph_active(const mg_model* m) { return m && m->graph_active ? 1 : 0; }
int mg_set_shared_gpu(mg_model* m, int shared) {
    if (!m) return fail(MG_E_ARG, "mg_set_shared_gpu: null model");
    std::lock_guard<std::recursive_mutex> lk(m->call_mu);
    const int prev = m->shared_gpu;
    m->shared_gpu = shared ? 1 : 0;
    if (m->shared_gpu) attention_step_allow_shared();
    if (prev != m->shared_gpu) { m->step_graph.reset(); m->stream_graph.reset(); }      // (the captured launches carry the LDS request)
    return prev;
}
int mg_set_cross_absorb(mg_model* m, int absorb, int key_splits) {
    if (!m) return fail(MG_E_ARG, "mg_set_cross_absorb: null model");
    if (key_splits < 0 || key_splits > 4) return fail(MG_E_ARG, "mg_set_cross_absorb: key_splits must be in [0, 4] (0 = keep)");
    std::lock_guard<std::recursive_mutex> lk(m->call_mu);
    const int prev = m->absorb;
    if (absorb < 0) return prev;                        // query
    if (absorb > 2) return fail(MG_E_ARG, "mg_set_cross_absorb: absorb must be 0 (K / V form), 1 (absorbed), 2 (by the call's rows) or < 0 (query)");
    if (absorb && !xattn_supported(m->d, m->H)) return fail(MG_E_UNSUPPORTED, "mg_set_cross_absorb: d_model %d / %d heads have no absorbed form", m->d, m->H);
    m->absorb = absorb;
    if (key_splits) m->xa_split = key_splits;
    m->step_graph.reset(); m->stream_graph.reset();      // (the captured steps hold the other form's launches and buffers)
    return prev;
}
// launches timed, their summed duration, and the summed number of (image, key) rows streamed per launch
int mg_profile_read(mg_model* m, long* launches, double* total_ms, double* total_keys) {
    if (!m) return fail(MG_E_ARG, "mg_profile_read: null model");
    if (launches) *launches = m->prof_n;
    if (total_ms) *total_ms = m->prof_ms;
    if (total_keys) *total_keys = m->prof_keys;
    return MG_OK;
}
// summed duration of the EMPTY event brackets (two hipEventRecord back to back) recorded after each timed launch:
// what the bracket itself costs on this stream, to be subtracted from total_ms for the kernel's own duration
int mg_profile_read_overhead(mg_model* m, double* empty_ms) {
    if (!m || !empty_ms) return fail(MG_E_ARG, "mg_profile_read_overhead: null argument");
    *empty_ms = m->prof_empty_ms;
    return MG_OK;
}

// Self-test of the hardware assumptions (MFMA 32x32x16 operand/accumulator layout, global_load_lds destination
// rule, cross-half exchange): an asymmetric 64x96x128 GEMM in both kernel shapes against a host fp64 reference.
int mg_selftest(void* stream, void* scratch_256k, char* msg_host, int msg_len) {
    if (!scratch_256k) return fail(MG_E_ARG, "mg_selftest: null scratch");
    mgStream_t st = (mgStream_t)stream;
    const int M = 64, N = 96, K = 128;
    std::vector<uint16_t> X(pk_elems(M, K)), W(pk_elems(N, K));
    std::vector<float> xf((size_t)M * K), wf((size_t)N * K), out((size_t)M * N), ref((size_t)M * N);
    auto tobf = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
    auto tof = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
    for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) { const uint16_t b = tobf(0.01f * (float)((m * 7 + k * 3) % 41) - 0.2f); X[pk_off(m, k, K)] = b; xf[(size_t)m * K + k] = tof(b); }
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) { const uint16_t b = tobf(0.02f * (float)((n * 5 + k * 11) % 29) - 0.3f); W[pk_off(n, k, K)] = b; wf[(size_t)n * K + k] = tof(b); }
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)xf[(size_t)m * K + k] * wf[(size_t)n * K + k]; ref[(size_t)m * N + n] = (float)s; }
    char* base = (char*)scratch_256k;
    uint16_t* dX = (uint16_t*)base;
    uint16_t* dW = (uint16_t*)(base + 32768);
    float* dO = (float*)(base + 65536);
    mg_memcpy_async(dX, X.data(), X.size() * 2, st);
    mg_memcpy_async(dW, W.data(), W.size() * 2, st);
    double worst = 0;
    for (int mode = 0; mode < 2; ++mode) {
        GemmArgs a = gemm_args(dX, dW, M, N, K);
        a.out_f32 = dO; a.ldo = N;
        mode == 0 ? gemm(a, EPI_F32_STORE, st) : gemm_rows(a, EPI_F32_STORE, st);
        mg_memcpy_async(out.data(), dO, out.size() * 4, st);
        mg_stream_sync(st);
        for (size_t i = 0; i < out.size(); ++i) { const double e = fabs((double)out[i] - ref[i]); if (e > worst) worst = e; }
    }
    const int rc = check_launch("mg_selftest");
    if (msg_host && msg_len > 0) snprintf(msg_host, msg_len, "gemm 64x96x128 tiled+rows max abs err %.3g (%s)", worst, worst < 1e-3 ? "ok" : "FAIL");
    if (rc != MG_OK) return rc;
    return worst < 1e-3 ? MG_OK : fail(MG_E_HIP, "mg_selftest: MFMA/LDS layout assumptions violated (max err %.3g)", worst);
}

int mg_beam_reorder(void* stream, const void* kv_src, void* kv_dst, const int32_t* beam_idx, int layers, int rows, int H,
                    int t_cap, int t_used) {
    if (!kv_src || !kv_dst || !beam_idx || t_used > t_cap) return fail(MG_E_ARG, "mg_beam_reorder: bad argument");
    beam_reorder_copy((const uint16_t*)kv_src, (uint16_t*)kv_dst, beam_idx, layers * 2, rows, H, t_cap, t_used, (mgStream_t)stream);
    return check_launch("mg_beam_reorder");
}

}  // extern "C"
