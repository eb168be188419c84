// Beam search on the device, restating stock transformers 5.15 `_beam_search` (generation/utils.py:3208-3525):
// log_softmax + running scores, top-2K over the K*V continuations (utils.py:3077-3130), running beams for the next
// step (3131-3152), finished-beam merge (3153-3206), early-stop heuristic (3008-3053), loop condition (3055-3075).
// The KV-cache reorder (cache_utils.py:100-104) exists in two forms: an ancestor table (the product path: the
// single-query attention kernel gathers each cached position from the physical row that wrote it, so no K/V bytes
// move) and a physical index_select copy (the reference's behaviour; exposed for the micro-benchmark).
// Selection order everywhere: value descending, then index ascending (deterministic).
#include "mg_kernels.h"

namespace mg {

struct BeamLayout {
    size_t running_seq, sequences, top_seq, tmp_seq, tmp_idx, running_scores, beam_scores, topv, topi, is_fin, heur, run_idx, beam_idx_out,
        top_run_idx, flags, rowv, rowi, total;
};
static BeamLayout beam_layout(int B, int K, int max_len) {
    BeamLayout l;
    size_t off = 0;
    auto take = [&](size_t bytes) { off = (off + 255) / 256 * 256; size_t o = off; off += bytes; return o; };
    const size_t ml = max_len, il = max_len - 1;
    l.running_seq = take((size_t)B * K * ml * 8);
    l.sequences = take((size_t)B * K * ml * 8);
    l.top_seq = take((size_t)B * 2 * K * ml * 8);
    l.tmp_seq = take((size_t)B * K * ml * 8);
    l.tmp_idx = take((size_t)B * K * ml * 4);
    l.running_scores = take((size_t)B * K * 4);
    l.beam_scores = take((size_t)B * K * 4);
    l.topv = take((size_t)B * 2 * K * 4);
    l.topi = take((size_t)B * 2 * K * 4);
    l.is_fin = take((size_t)B * K);
    l.heur = take((size_t)B);
    l.run_idx = take((size_t)B * K * il * 4);
    l.beam_idx_out = take((size_t)B * K * il * 4);
    l.top_run_idx = take((size_t)B * 2 * K * il * 4);
    l.flags = take((size_t)B * 4 * 4);
    l.rowv = take((size_t)B * K * 2 * K * 4);      // per (image, beam) row: its own top-2K candidates
    l.rowi = take((size_t)B * K * 2 * K * 4);
    l.total = (off + 255) / 256 * 256;
    return l;
}
size_t beam_state_bytes(int B, int K, int max_len) { return beam_layout(B, K, max_len).total; }

struct BeamPtrs {
    int64_t *running_seq, *sequences, *top_seq, *tmp_seq;
    int* tmp_idx;
    float *running_scores, *beam_scores, *topv;
    int* topi;
    uint8_t *is_fin, *heur;
    int *run_idx, *beam_idx_out, *top_run_idx, *flags;
    float* rowv;
    int* rowi;
};
static BeamPtrs beam_ptrs(void* state, int B, int K, int max_len) {
    const BeamLayout l = beam_layout(B, K, max_len);
    char* s = (char*)state;
    BeamPtrs p;
    p.running_seq = (int64_t*)(s + l.running_seq); p.sequences = (int64_t*)(s + l.sequences); p.top_seq = (int64_t*)(s + l.top_seq);
    p.tmp_seq = (int64_t*)(s + l.tmp_seq); p.tmp_idx = (int*)(s + l.tmp_idx);
    p.running_scores = (float*)(s + l.running_scores); p.beam_scores = (float*)(s + l.beam_scores); p.topv = (float*)(s + l.topv);
    p.topi = (int*)(s + l.topi); p.is_fin = (uint8_t*)(s + l.is_fin); p.heur = (uint8_t*)(s + l.heur);
    p.run_idx = (int*)(s + l.run_idx); p.beam_idx_out = (int*)(s + l.beam_idx_out); p.top_run_idx = (int*)(s + l.top_run_idx);
    p.flags = (int*)(s + l.flags); p.rowv = (float*)(s + l.rowv); p.rowi = (int*)(s + l.rowi);
    return p;
}

// utils.py:3316-3351 initial values (fill value = pad_token_id or eos: pad id 0 is falsy -> EOS, utils.py:3319)
__global__ __launch_bounds__(256) void beam_init_kernel(BeamPtrs p, int B, int K, int max_len, int fill, int start, int64_t* next_ids,
                                                   int* anc, int T_cap, int* counters) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int R = B * K;
    for (int i = tid; i < K * max_len; i += 256) {
        const int64_t v = (i % max_len == 0) ? start : fill;
        p.running_seq[(size_t)b * K * max_len + i] = v;
        p.sequences[(size_t)b * K * max_len + i] = v;
    }
    for (int i = tid; i < K * (max_len - 1); i += 256) {
        p.run_idx[(size_t)b * K * (max_len - 1) + i] = -1;
        p.beam_idx_out[(size_t)b * K * (max_len - 1) + i] = -1;
    }
    for (int i = tid; i < K; i += 256) {
        p.running_scores[b * K + i] = i == 0 ? 0.f : -1.0e9f;
        p.beam_scores[b * K + i] = -1.0e9f;
        p.is_fin[b * K + i] = 0;
        next_ids[b * K + i] = start;
    }
    for (int i = tid; i < T_cap * K; i += 256) {   // identity ancestor table for this image's rows
        const int j = i / K, k = i - j * K;
        anc[(size_t)j * R + b * K + k] = b * K + k;
    }
    if (tid == 0) {
        p.heur[b] = 1;
        if (b == 0) { counters[0] = 1; counters[1] = -1; counters[2] = 0; counters[4] = 0; }
    }
}
void beam_init(void* state, int B, int K, int max_len, int pad, int eos, int start, int64_t* next_ids, int* anc, int T_cap,
               int* counters, mgStream_t stream) {
    const BeamPtrs p = beam_ptrs(state, B, K, max_len);
    MG_LAUNCH(beam_init_kernel, dim3(B), dim3(256), 0, stream, p, B, K, max_len, pad ? pad : eos, start, next_ids, anc, T_cap, counters);
}

MG_DEV bool cand_before(float v1, int i1, float v2, int i2) { return v1 > v2 || (v1 == v2 && i1 < i2); }

// a. log-probs + running scores (utils.py:3388-3396), b. top-2K over K*V (utils.py:3077-3130), in two steps:
//  1. one workgroup of 1024 threads per (image, beam) row: log-softmax statistics, then that row's own top-2K by 2K
//     rounds of block-wide selection over values held in registers (at most 2K of the image's top-2K come from one row);
//  2. one workgroup per image ranks the K*2K row candidates (value descending, flat index ascending - torch.topk's
//     order for distinct values) and keeps the first 2K.
constexpr int BT_THREADS = 1024, BT_NPT = 40;     // register-resident path for V <= 40960, generic loop beyond
MG_DEV float block_max16(float v, float* red, int tid) {
    v = wave_max(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < BT_THREADS / 64; ++i) r = fmaxf(r, red[i]);
    __syncthreads();
    return r;
}
MG_DEV float block_sum16(float v, float* red, int tid) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < BT_THREADS / 64; ++i) r += red[i];
    __syncthreads();
    return r;
}
__global__ __launch_bounds__(BT_THREADS) void beam_row_topk_kernel(BeamPtrs p, const float* logits, int ldl, int V, int K, int cur_len_arg,
                                                              int eos, int min_len, const int* counters, const int* tdev, BeamSlots bs) {
    // queue form (bs.pos): every image slot is at its own length, idle slots are skipped; batch form: one length, the batch's
    // continue flag
    if (bs.pos ? bs.live[blockIdx.x] == 0 : counters[0] == 0) return;
    const int cur_len = bs.pos ? bs.pos[blockIdx.x] + 1 : (tdev ? *tdev + 1 : cur_len_arg);
    MG_DYN_SMEM(smem);
    float* red = (float*)smem;            // [16]
    float* selv = red + 16;               // [16]
    int* seli = (int*)(selv + 16);        // [16]
    const int row = blockIdx.x, k = row % K, tid = threadIdx.x;
    const bool no_eos = cur_len < min_len;
    const float* lg = logits + (size_t)row * ldl;
    const bool in_regs = V <= BT_THREADS * BT_NPT;
    float x[BT_NPT];
    float mx = -3.0e38f;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < BT_NPT; ++j) {
            const int v = tid + j * BT_THREADS;
            x[j] = v < V ? lg[v] : -3.0e38f;
            mx = fmaxf(mx, x[j]);
        }
    } else {
        for (int v = tid; v < V; v += BT_THREADS) mx = fmaxf(mx, lg[v]);
    }
    mx = block_max16(mx, red, tid);
    float s = 0.f;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < BT_NPT; ++j) s += (tid + j * BT_THREADS < V) ? expf(x[j] - mx) : 0.f;
    } else {
        for (int v = tid; v < V; v += BT_THREADS) s += expf(lg[v] - mx);
    }
    s = block_sum16(s, red, tid);
    const float lse = logf(s), rs = p.running_scores[row];
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < BT_NPT; ++j) {
            const int v = tid + j * BT_THREADS;
            float lp = (x[j] - mx) - lse;
            if (no_eos && v == eos) lp = -INFINITY;
            x[j] = v < V ? lp + rs : -INFINITY;
        }
    }
    float lastv = 3.0e38f;
    int lasti = -1;
    const int keep = 2 * K;
    for (int round = 0; round < keep; ++round) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < BT_NPT; ++j) {
                const int v = tid + j * BT_THREADS;
                const int idx = k * V + v;
                if (v < V && cand_before(lastv, lasti, x[j], idx) && cand_before(x[j], idx, bv, bi)) { bv = x[j]; bi = idx; }
            }
        } else {
            for (int v = tid; v < V; v += BT_THREADS) {
                float lp = (lg[v] - mx) - lse;
                if (no_eos && v == eos) lp = -INFINITY;
                const float val = lp + rs;
                const int idx = k * V + v;
                if (cand_before(lastv, lasti, val, idx) && cand_before(val, idx, bv, bi)) { bv = val; bi = idx; }
            }
        }
#pragma unroll
        for (int step = 1; step < 64; step <<= 1) {
            const float ov = __shfl_xor(bv, step);
            const int oi = __shfl_xor(bi, step);
            if (cand_before(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { selv[tid >> 6] = bv; seli[tid >> 6] = bi; }
        __syncthreads();
        for (int ww = 0; ww < BT_THREADS / 64; ++ww)
            if (cand_before(selv[ww], seli[ww], bv, bi)) { bv = selv[ww]; bi = seli[ww]; }
        __syncthreads();
        lastv = bv; lasti = bi;
        if (tid == 0) { p.rowv[(size_t)row * keep + round] = bv; p.rowi[(size_t)row * keep + round] = bi; }
    }
}
// rank of every row candidate among the image's K*2K; ranks < 2K are the image's top-2K in order
__global__ __launch_bounds__(128) void beam_merge_topk_kernel(BeamPtrs p, int K, const int* counters, BeamSlots bs) {
    if (bs.pos ? bs.live[blockIdx.x * K] == 0 : counters[0] == 0) return;
    MG_DYN_SMEM(smem);
    const int b = blockIdx.x, tid = threadIdx.x, keep = 2 * K, n = K * keep;
    float* cv = (float*)smem;             // [128]
    int* ci = (int*)(cv + 128);           // [128]
    if (tid < n) { cv[tid] = p.rowv[(size_t)b * n + tid]; ci[tid] = p.rowi[(size_t)b * n + tid]; }
    __syncthreads();
    if (tid < n) {
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += cand_before(cv[j], ci[j], cv[tid], ci[tid]) ? 1 : 0;
        if (rank < keep) { p.topv[b * keep + rank] = cv[tid]; p.topi[b * keep + rank] = ci[tid]; }
    }
}

// c.-g. bookkeeping for one image (K <= 8, 2K <= 16 candidates)
__global__ __launch_bounds__(256) void beam_update_kernel(BeamPtrs p, int B, int K, int V, int max_len, int cur_len_arg, int eos,
                                                     float div_arg, const float* div_table, int early_stopping, int64_t* next_ids,
                                                     int* beam_idx, const int* counters, const int* tdev, BeamSlots bs) {
    if (bs.pos ? bs.live[blockIdx.x * K] == 0 : counters[0] == 0) return;
    const int cur_len = bs.pos ? bs.pos[blockIdx.x * K] + 1 : (tdev ? *tdev + 1 : cur_len_arg);
    // (cur_len + 1 - prompt_len)^length_penalty with prompt_len = 1: the divisor of the finished-beam score
    // (utils.py:3182) and, after the increment of cur_len, of the early-stop heuristic (utils.py:3047-3052)
    const float fin_div = div_table ? div_table[cur_len] : div_arg;
    const float heur_div = fin_div;
    MG_DYN_SMEM(smem);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int keep = 2 * K, ml = max_len, il = max_len - 1;
    // Only columns <= cur_len are live: every row of the sequence arrays holds the fill value (the index histories -1) beyond the
    // column written by the latest step, in every slot, so the gathers below move columns [0, cur_len] ([0, cur_len) for the
    // histories) and leave the tails as they are - the same arrays as a copy of whole rows, at cur_len / max_len of the traffic.
    const int mle = cur_len + 1 < ml ? cur_len + 1 : ml, ile = cur_len < il ? cur_len : il;
    int* src = (int*)smem;                  // [16] source beam of candidate c
    int* hit = src + 16;                    // [16]
    int* nxt = hit + 16;                    // [8] candidate chosen as running beam k
    int* sel = nxt + 8;                     // [8] merged index chosen as finished slot k
    float* fin = (float*)(sel + 8);         // [16]
    float* runlp = fin + 16;                // [16]
    float* msc = runlp + 16;                // [24]
    int* mfin = (int*)(msc + 24);           // [24]
    int* misc = mfin + 24;                  // [4]
    int64_t* rs = p.running_seq + (size_t)b * K * ml;
    int64_t* sq = p.sequences + (size_t)b * K * ml;
    int64_t* ts = p.top_seq + (size_t)b * keep * ml;
    int* ri = p.run_idx + (size_t)b * K * il;
    int* bo = p.beam_idx_out + (size_t)b * K * il;
    int* tr = p.top_run_idx + (size_t)b * keep * il;

    if (tid < keep) {
        const int idx = p.topi[b * keep + tid];
        src[tid] = idx / V;
        const int tok = idx - src[tid] * V;
        hit[tid] = (tok == eos) || (cur_len + 1 >= max_len);
    }
    __syncthreads();
    // top-2K sequences / beam-index histories (utils.py:3115-3127)
    for (int i = tid; i < keep * mle; i += 256) {
        const int c = i / mle, j = i - c * mle;
        int64_t v = rs[(size_t)src[c] * ml + j];
        if (j == cur_len) v = p.topi[b * keep + c] - src[c] * V;
        ts[(size_t)c * ml + j] = v;
    }
    for (int i = tid; i < keep * ile; i += 256) {
        const int c = i / ile, j = i - c * ile;
        int v = ri[(size_t)src[c] * il + j];
        if (j == cur_len - 1) v = src[c] + b * K;
        tr[(size_t)c * il + j] = v;
    }
    if (tid < keep) {
        const float tv = p.topv[b * keep + tid];
        runlp[tid] = tv + (hit[tid] ? 1.0f : 0.0f) * -1.0e9f;            // utils.py:3145
        bool allfin = true;
        for (int k = 0; k < K; ++k) allfin = allfin && p.is_fin[b * K + k];
        float f = tv / fin_div;                                           // utils.py:3182
        f += ((allfin && early_stopping) ? 1.0f : 0.0f) * -1.0e9f;        // 3184-3185
        f += (p.heur[b] ? 0.0f : 1.0f) * -1.0e9f;                         // 3187
        const bool just = hit[tid] && tid < K;                            // 3178
        f += (just ? 0.0f : 1.0f) * -1.0e9f;                              // 3190
        fin[tid] = f;
    }
    __syncthreads();
    if (tid == 0) {
        // e. running beams: top-K of runlp (utils.py:3147)
        bool used[16];
        for (int c = 0; c < keep; ++c) used[c] = false;
        for (int k = 0; k < K; ++k) {
            int best = -1;
            for (int c = 0; c < keep; ++c)
                if (!used[c] && (best < 0 || runlp[c] > runlp[best])) best = c;
            used[best] = true;
            nxt[k] = best;
        }
        // f. finished beams: top-K of [beam_scores (K) | fin (2K)] (utils.py:3195-3203)
        for (int k = 0; k < K; ++k) { msc[k] = p.beam_scores[b * K + k]; mfin[k] = p.is_fin[b * K + k]; }
        for (int c = 0; c < keep; ++c) { msc[K + c] = fin[c]; mfin[K + c] = hit[c] && c < K; }
        bool used2[24];
        for (int c = 0; c < K + keep; ++c) used2[c] = false;
        for (int k = 0; k < K; ++k) {
            int best = -1;
            for (int c = 0; c < K + keep; ++c)
                if (!used2[c] && (best < 0 || msc[c] > msc[best])) best = c;
            used2[best] = true;
            sel[k] = best;
        }
    }
    __syncthreads();
    // new finished set: gathered from the OLD sequences / candidates into temporaries, then copied back
    int64_t* tq = p.tmp_seq + (size_t)b * K * ml;
    int* ti = p.tmp_idx + (size_t)b * K * ml;
    for (int i = tid; i < K * mle; i += 256) {
        const int k = i / mle, j = i - k * mle, s2 = sel[k];
        tq[(size_t)k * ml + j] = s2 < K ? sq[(size_t)s2 * ml + j] : ts[(size_t)(s2 - K) * ml + j];
    }
    for (int i = tid; i < K * ile; i += 256) {
        const int k = i / ile, j = i - k * ile, s2 = sel[k];
        ti[(size_t)k * il + j] = s2 < K ? bo[(size_t)s2 * il + j] : tr[(size_t)(s2 - K) * il + j];
    }
    __syncthreads();
    for (int i = tid; i < K * mle; i += 256) { const int k = i / mle, j = i - k * mle; sq[(size_t)k * ml + j] = tq[(size_t)k * ml + j]; }
    for (int i = tid; i < K * ile; i += 256) { const int k = i / ile, j = i - k * ile; bo[(size_t)k * il + j] = ti[(size_t)k * il + j]; }
    for (int i = tid; i < K * mle; i += 256) { const int k = i / mle, j = i - k * mle; rs[(size_t)k * ml + j] = ts[(size_t)nxt[k] * ml + j]; }
    for (int i = tid; i < K * ile; i += 256) { const int k = i / ile, j = i - k * ile; ri[(size_t)k * il + j] = tr[(size_t)nxt[k] * il + j]; }
    __syncthreads();
    if (tid == 0) {
        float nbs[8], nrs[8];
        int nf[8];
        for (int k = 0; k < K; ++k) { nbs[k] = msc[sel[k]]; nf[k] = mfin[sel[k]]; nrs[k] = runlp[nxt[k]]; }
        bool allfin = true, allhit = true;
        for (int k = 0; k < K; ++k) {
            p.beam_scores[b * K + k] = nbs[k];
            p.is_fin[b * K + k] = (uint8_t)nf[k];
            p.running_scores[b * K + k] = nrs[k];
            allfin = allfin && nf[k];
        }
        for (int c = 0; c < keep; ++c) allhit = allhit && hit[c];
        // g. early-stop heuristic with the incremented cur_len (utils.py:3047-3052)
        float worst = nbs[0];
        for (int k = 1; k < K; ++k) worst = fminf(worst, nbs[k]);
        const float best_possible = nrs[0] / heur_div;
        bool any = false;
        for (int k = 0; k < K; ++k) any = any || (best_possible > (nf[k] ? worst : -1.0e9f));
        const bool h = p.heur[b] && any;
        p.heur[b] = h;
        p.flags[b * 4 + 0] = h; p.flags[b * 4 + 1] = allfin; p.flags[b * 4 + 2] = allhit;
    }
    __syncthreads();
    if (tid < K) {
        next_ids[b * K + tid] = rs[(size_t)tid * ml + cur_len];
        beam_idx[b * K + tid] = ri[(size_t)tid * il + cur_len - 1];
    }
}

// loop condition (utils.py:3055-3075), reduced over the batch; counters[0] = continue?
__global__ void beam_flags_kernel(BeamPtrs p, int B, int early_stopping, int* counters) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || counters[0] == 0) return;
    bool any_h = false, all_fin = true, all_hit = true;
    for (int b = 0; b < B; ++b) {
        any_h = any_h || p.flags[b * 4 + 0];
        all_fin = all_fin && p.flags[b * 4 + 1];
        all_hit = all_hit && p.flags[b * 4 + 2];
    }
    counters[0] = (any_h && !(all_fin && early_stopping) && !all_hit) ? 1 : 0;
}

float beam_length_divisor(int cur_len, float length_penalty) { return (float)pow((double)cur_len, (double)length_penalty); }

// tdev != nullptr (graph capture): cur_len = *tdev + 1 and the divisor comes from div_table[cur_len] (device memory,
// filled on the host with beam_length_divisor so both forms use bit-identical values)
void beam_step(void* state, const float* logits, int ldl, int V, int B, int K, int max_len, int cur_len, const int* tdev,
               const float* div_table, int eos, int min_len, float length_penalty, int early_stopping, int64_t* next_ids,
               int* beam_idx, int* counters, mgStream_t stream, const BeamSlots* slots) {
    const BeamPtrs p = beam_ptrs(state, B, K, max_len);
    const BeamSlots bs = slots ? *slots : BeamSlots{nullptr, nullptr};
    MG_LAUNCH(beam_row_topk_kernel, dim3(B * K), dim3(BT_THREADS), 256, stream, p, logits, ldl, V, K, cur_len, eos, min_len, (const int*)counters, tdev, bs);
    MG_LAUNCH(beam_merge_topk_kernel, dim3(B), dim3(128), 1024, stream, p, K, (const int*)counters, bs);
    const float div = beam_length_divisor(cur_len, length_penalty);
    MG_LAUNCH(beam_update_kernel, dim3(B), dim3(256), 1024, stream, p, B, K, V, max_len, cur_len, eos, div, (tdev || bs.pos) ? div_table : nullptr,
              early_stopping, next_ids, beam_idx, (const int*)counters, tdev, bs);
    if (!bs.pos) MG_LAUNCH(beam_flags_kernel, dim3(1), dim3(64), 0, stream, p, B, early_stopping, counters);
}

// utils.py:3510-3523: best beam per image, generated length from its beam-index history
__global__ __launch_bounds__(256) void beam_finalize_kernel(BeamPtrs p, int B, int K, int max_len, int64_t* out_ids, int* out_cols,
                                                       float* out_scores) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int ml = max_len, il = max_len - 1;
    for (int j = tid; j < ml; j += 256) out_ids[(size_t)b * ml + j] = p.sequences[(size_t)b * K * ml + j];
    int c = 0;
    for (int j = tid; j < il; j += 256) c += p.beam_idx_out[(size_t)b * K * il + j] != -1;
    MG_DYN_SMEM(smem);
    int* tot = (int*)smem;
    if (tid == 0) *tot = 0;
    __syncthreads();
    if (c) atomicAdd(tot, c);
    __syncthreads();
    if (tid == 0) {
        atomicMax(out_cols, 1 + *tot);
        if (out_scores) out_scores[b] = p.beam_scores[b * K];
    }
}
void beam_finalize(void* state, int B, int K, int max_len, int64_t* out_ids, int* out_cols, float* out_scores, mgStream_t stream) {
    const BeamPtrs p = beam_ptrs(state, B, K, max_len);
    mg_memset_async(out_cols, 0, sizeof(int), stream);
    MG_LAUNCH(beam_finalize_kernel, dim3(B), dim3(256), 16, stream, p, B, K, max_len, out_ids, out_cols, out_scores);
}

// ancestor-table reorder: one workgroup per cached position j < t_written, in place
__global__ __launch_bounds__(256) void beam_reorder_anc_kernel(int* anc, const int* beam_idx, int rows, const int* counters,
                                                          const int* tdev, BeamSlots bs) {
    const int j = blockIdx.x;
    if (!bs.pos) {
        if (counters[0] == 0) return;
        if (tdev && j > *tdev) return;      // graph form: launched over every position, only the written ones are permuted
    }
    int v[4];
    int n = 0;
    // queue form: a row is permuted at position j when its slot is live and has written that position (rows of other slots keep theirs)
    for (int r = threadIdx.x; r < rows; r += 256) {
        const bool on = !bs.pos || (bs.live[r] != 0 && j <= bs.pos[r]);
        v[n++] = anc[(size_t)j * rows + (on ? beam_idx[r] : r)];
    }
    __syncthreads();
    n = 0;
    for (int r = threadIdx.x; r < rows; r += 256) anc[(size_t)j * rows + r] = v[n++];
}
void beam_reorder_anc(int* anc, const int* beam_idx, int rows, int t_written, const int* tdev, const int* counters, mgStream_t stream,
                      const BeamSlots* slots) {
    const BeamSlots bs = slots ? *slots : BeamSlots{nullptr, nullptr};
    MG_LAUNCH(beam_reorder_anc_kernel, dim3(t_written), dim3(256), 0, stream, anc, beam_idx, rows, counters, tdev, bs);
}

// ---------------------------------------------------------------------------------------------------------
// Queue form (mg_generate_stream_beam): the image slots of a continuous beam decoder.  A slot = the K rows of one image; every slot
// runs the batch form's per-image arithmetic at its own length (BeamSlots), so an image's hypotheses are what a batch call returns
// for it: in the batch form an image whose own stopping condition holds is frozen (its heuristic flag gates every new finished
// candidate with -1e9, utils.py:3187) until the last image of the batch stops.
//   end    per live slot: the image's own loop condition (utils.py:3055-3075 without the reduction over the batch); a stopped image
//          is written out (best hypothesis, its length from the beam-index history, its score) and the slot becomes idle,
//          otherwise the slot's rows advance one position
//   assign one thread walks the slots in order and hands idle ones the next queued images whose cross K/V are in the pool
//   init   per newly assigned slot: the batch form's initial values (beam_init_kernel), identity ancestors, start tokens
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void beam_slot_end_kernel(BeamPtrs p, int K, int max_len, int early_stopping, int* pos, int* img, int* live,
                                                       int64_t* out_ids, int* out_len, float* out_scores, int* ctr) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (live[b * K] == 0) return;
    const bool cont = p.flags[b * 4 + 0] && !(p.flags[b * 4 + 1] && early_stopping) && !p.flags[b * 4 + 2];
    if (cont) {
        if (tid < K) pos[b * K + tid] += 1;
        return;
    }
    const int i = img[b * K];
    const int ml = max_len, il = max_len - 1;
    for (int j = tid; j < ml; j += 256) out_ids[(size_t)i * ml + j] = p.sequences[(size_t)b * K * ml + j];
    int c = 0;
    for (int j = tid; j < il; j += 256) c += p.beam_idx_out[(size_t)b * K * il + j] != -1;
    MG_DYN_SMEM(smem);
    int* tot = (int*)smem;
    if (tid == 0) *tot = 0;
    __syncthreads();
    if (c) atomicAdd(tot, c);
    __syncthreads();
    if (tid == 0) {
        out_len[i] = 1 + *tot;
        if (out_scores) out_scores[i] = p.beam_scores[b * K];
        atomicAdd(ctr + 1, 1);
    }
    __syncthreads();
    if (tid < K) { live[b * K + tid] = 0; img[b * K + tid] = -1; }
}
__global__ __launch_bounds__(64) void beam_slot_assign_kernel(int slots, int K, const int* live, const int* img, int* assign, int* ctr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int head = ctr[4];
    const int ready = ctr[5];
    int n_live = 0, oldest = 0x7fffffff;
    for (int b = 0; b < slots; ++b) {
        int a = -1, im = img[b * K];
        if (live[b * K] == 0 && head < ready) { a = head++; im = a; }
        assign[b] = a;
        if (live[b * K] != 0 || a >= 0) { ++n_live; oldest = im < oldest ? im : oldest; }
    }
    ctr[4] = head;
    ctr[0] = n_live;
    ctr[7] = n_live ? oldest : head;      // every image below this index has finished
    ctr[2] += 1;
}
__global__ __launch_bounds__(256) void beam_slot_init_kernel(BeamPtrs p, int slots, int K, int max_len, int fill, int start, const int* assign,
                                                        int* pos, int* img, int* pool, int* bpool, int* live, int64_t* next_ids, int* anc,
                                                        int T_cap, int pool_cap) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int a = assign[b];
    if (a < 0) return;
    const int R = slots * K;
    for (int i = tid; i < K * max_len; i += 256) {
        const int64_t v = (i % max_len == 0) ? start : fill;
        p.running_seq[(size_t)b * K * max_len + i] = v;
        p.sequences[(size_t)b * K * max_len + i] = v;
    }
    for (int i = tid; i < K * (max_len - 1); i += 256) {
        p.run_idx[(size_t)b * K * (max_len - 1) + i] = -1;
        p.beam_idx_out[(size_t)b * K * (max_len - 1) + i] = -1;
    }
    for (int i = tid; i < K; i += 256) {
        p.running_scores[b * K + i] = i == 0 ? 0.f : -1.0e9f;
        p.beam_scores[b * K + i] = -1.0e9f;
        p.is_fin[b * K + i] = 0;
        next_ids[b * K + i] = start;
        pos[b * K + i] = 0; img[b * K + i] = a; pool[b * K + i] = a % pool_cap; live[b * K + i] = 1;
    }
    for (int i = tid; i < T_cap * K; i += 256) {   // identity ancestor table for this slot's rows
        const int j = i / K, k = i - j * K;
        anc[(size_t)j * R + b * K + k] = b * K + k;
    }
    if (tid == 0) {
        p.heur[b] = 1;
        bpool[b] = a % pool_cap;
        p.flags[b * 4 + 0] = 1; p.flags[b * 4 + 1] = 0; p.flags[b * 4 + 2] = 0;
    }
}
void beam_slots_step(void* state, int slots, int K, int max_len, int pad, int eos, int start, int early_stopping, int* pos, int* img, int* pool,
                     int* bpool, int* live, int* assign, int64_t* next_ids, int* anc, int T_cap, int pool_cap, int64_t* out_ids, int* out_len,
                     float* out_scores, int* ctr, bool end_first, mgStream_t stream) {
    const BeamPtrs p = beam_ptrs(state, slots, K, max_len);
    if (end_first)
        MG_LAUNCH(beam_slot_end_kernel, dim3(slots), dim3(256), 16, stream, p, K, max_len, early_stopping, pos, img, live, out_ids, out_len, out_scores, ctr);
    MG_LAUNCH(beam_slot_assign_kernel, dim3(1), dim3(64), 0, stream, slots, K, (const int*)live, (const int*)img, assign, ctr);
    MG_LAUNCH(beam_slot_init_kernel, dim3(slots), dim3(256), 0, stream, p, slots, K, max_len, pad ? pad : eos, start, (const int*)assign, pos, img,
              pool, bpool, live, next_ids, anc, T_cap, pool_cap);
}

// physical reorder (cache_utils.py:100-104): dst[lk][r] = src[lk][beam_idx[r]], 16-byte copies, HBM-bound
__global__ __launch_bounds__(256) void beam_reorder_copy_kernel(const uint4* src, uint4* dst, const int* beam_idx, int rows, int H,
                                                           int t_cap, int t_used) {
    const size_t per_h = (size_t)t_cap * 8, used_h = (size_t)t_used * 8;      // uint4 per (row, head)
    const size_t per_row = per_h * H;
    const int lk = blockIdx.y;
    const size_t n = (size_t)rows * H * used_h;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (H * used_h), rem = i - r * (H * used_h);
        const size_t h = rem / used_h, e = rem - h * used_h;
        const size_t so = ((size_t)lk * rows + beam_idx[r]) * per_row + h * per_h + e;
        const size_t dof = ((size_t)lk * rows + r) * per_row + h * per_h + e;
        dst[dof] = src[so];
    }
}
void beam_reorder_copy(const uint16_t* src, uint16_t* dst, const int* beam_idx, int nlk, int rows, int H, int t_cap, int t_used,
                       mgStream_t stream) {
    const size_t n = (size_t)rows * H * t_used * 8;
    int bx = (int)((n + 255) / 256);
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    MG_LAUNCH(beam_reorder_copy_kernel, dim3(bx, nlk), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, beam_idx, rows, H, t_cap, t_used);
}

}  // namespace mg
