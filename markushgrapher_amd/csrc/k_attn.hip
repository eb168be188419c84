// Attention over packed Q / K / V^T with the relative-position bias fused into the score tile
// (the [B,H,S,S] bias of stock:904-953,1012-1029 is never materialised):
//   ATT_ENC      joint patch+text+layout self-attention: bias = 1-D(j-i) + horizontal + vertical buckets
//                (stock:956-1009; box centres in float64, (c_j-c_i) -> fp32, *100, truncation: stock:914-923)
//   ATT_DEC_SELF teacher-forced decoder self-attention: causal, T5 1-D bias of block 0 (stock:470-485,1234-1237)
//   ATT_CROSS    teacher-forced cross-attention: no bias (stock:543-550)
// No 1/sqrt(d_k) scaling (stock:402-403).  Masked keys get a large negative finite score.
//
// Structure: workgroup = 4 waves = 128 consecutive queries of one (b,h); each wave owns 32 queries.
// Scores are computed TRANSPOSED (S^T = K·Q^T, MFMA A = K fragment, B = Q fragment) so a lane owns ONE query:
// running max / sum / rescale are lane-local, and the bf16 P^T tile is directly the B operand of
// O^T = V^T · P^T after one exchange with lane^32.  K and V^T tiles of 64 keys are staged with global_load_lds
// (double-buffered, fragment order, conflict-free b128 reads); per-key metadata (mask, box centres) and the
// three bias tables of head h live in LDS for the whole workgroup.
#include "mg_kernels.h"
#include <atomic>

namespace mg {

constexpr int AT_KEYS = 64;                         // keys per stage
constexpr int AT_STAGE_BYTES = 16 * TILE_BYTES;     // 8 K fragments + 8 V^T fragments
constexpr float AT_NEG = -1.0e30f;

template <int MODE>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
    MG_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int nqb = (a.Sq_cap + 127) / 128;
    // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  All heads and query
    // blocks of one image go to the same XCD, so an image's per-(query, key) bucket-index array (2 B per pair, shared by
    // the 16 heads) and its K/V tiles are fetched into ONE L2 instead of eight (PMC: 0.97 GB fetched per launch
    // before, mostly that array eight times over).
    int bid = blockIdx.x;
    if ((a.B & 7) == 0) {
        const int per_img = a.H * nqb, xcd = bid & 7, j = bid >> 3;
        bid = ((j / per_img) * 8 + xcd) * per_img + (j % per_img);
    }
    const int qb = bid % nqb;
    const int bh = bid / nqb;
    const int h = bh % a.H, b = bh / a.H;
    const int Sk_pad = (a.Sk + AT_KEYS - 1) / AT_KEYS * AT_KEYS;
    // padded positions are never attended and their own outputs are never used: query blocks without an attended
    // position only clear their context rows (later GEMMs must see finite values), key stages without an attended key
    // are not visited
    const int* kst = (MODE == ATT_ENC && a.kst) ? a.kst + (size_t)b * (1 + (a.Sk_cap >> 6)) : nullptr;
    if (MODE == ATT_ENC && a.qbv && !a.qbv[(size_t)b * nqb + qb]) {
        const int HDz = a.H * 64;
        for (int i = tid; i < 128 * 8; i += 256) {
            const int q = qb * 128 + (i >> 3), c = (i & 7) * 8;
            if (q < a.Sq_cap) st16(a.ctx + pk_off(b * a.Sq_cap + q, h * 64 + c, HDz), make_uint4(0, 0, 0, 0));
        }
        return;
    }

    // LDS carve: [2 stages][tables][key mask bytes (decoder modes)]
    //   ATT_ENC:      t1[32], th[32], tv[32] = the three bucket tables of head h
    //   ATT_DEC_SELF: tab1[tab1_len] = causal T5 table of head h by distance
    char* st_base = smem;
    float* tab1 = (float*)(smem + 2 * AT_STAGE_BYTES);
    // ATT_ENC: t1 has 64 entries - the mask bit of a bucket-index word sits right above its 5-bit 1-D bucket, so
    // (e >> 10) & 63 selects entries 32..63 = AT_NEG for masked keys and the mask costs no instructions
    const int t1n = (MODE == ATT_CROSS) ? 0 : (MODE == ATT_ENC ? 64 : a.tab1_len);
    float* thv = tab1 + ((t1n + 3) & ~3);      // ATT_ENC: th[32] | tv[32] (32-entry tables: one entry per LDS bank, no conflicts)
    unsigned char* kmk = (unsigned char*)(thv + (MODE == ATT_ENC ? 64 : 0));

    if (MODE == ATT_ENC) {
        for (int i = tid; i < 64; i += 256) tab1[i] = i < 32 ? a.tab1[(size_t)i * a.H + h] : AT_NEG;
        for (int i = tid; i < 64; i += 256) thv[i] = i < 32 ? a.tabh[(size_t)i * a.H + h] : a.tabv[(size_t)(i - 32) * a.H + h];
    } else {
        for (int i = tid; i < t1n; i += 256) tab1[i] = a.tab1[(size_t)i * a.H + h];
        for (int i = tid; i < Sk_pad; i += 256)
            kmk[i] = (i < a.Sk) ? (a.kmask ? a.kmask[(size_t)b * a.Sk_cap + i] : 1) : 0;
    }

    // this wave's 32 queries
    const int q0 = qb * 128 + w * 32;
    int qrt = q0 >> 5;
    const int qrt_max = (a.Sq_cap >> 5) - 1;
    if (qrt > qrt_max) qrt = qrt_max;
    const uint16_t* Qb = a.Q + (((size_t)b * a.H + h) * (size_t)(a.Sq_cap >> 5) + (size_t)qrt) * (4 * TILE_ELEMS);
    uint4 qf[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) qf[kt] = ld16((const char*)(Qb + kt * TILE_ELEMS) + lane * 16);
    const int qi = q0 + l32;                 // this lane's query index
    // ATT_ENC: per-(query, key) bucket indices precomputed once per batch (bias_index_kernel), 16 keys = 32 B per lane
    // and 32-key tile, in accumulator-register order; prefetched one stage ahead
    const int qcl = qi < a.Sk_cap ? qi : a.Sk_cap - 1;
    const uint16_t* bix = (MODE == ATT_ENC)
        ? a.bidx + ((size_t)b * (size_t)(a.Sk_cap >> 5) * (size_t)a.Sk_cap + (size_t)qcl) * 32 + half * 16 : nullptr;
    const size_t bix_tile = (size_t)a.Sk_cap * 32;
    uint4 bcur[4], bnxt[4];
    auto load_bidx = [&](int st, uint4 (&d)[4]) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const uint16_t* p = bix + (size_t)(st * 2 + t2) * bix_tile;
            d[t2 * 2] = ld16(p);
            d[t2 * 2 + 1] = ld16(p + 8);
        }
    };
    const uint16_t* Kb = a.K + ((size_t)b * a.H + h) * (size_t)(a.Sk_cap >> 5) * (4 * TILE_ELEMS);
    const uint16_t* Vb = a.Vt + ((size_t)b * a.H + h) * 2 * (size_t)(a.Sk_cap >> 4) * TILE_ELEMS;
    const int krt_max = (a.Sk_cap >> 5) - 1, vkt_max = (a.Sk_cap >> 4) - 1;
    // stage loader: wave w copies fragments 4w..4w+3 (0..7 = K: key-tile f/4, dk-tile f%4; 8..15 = V^T: d-tile, key-k-tile)
    auto stage = [&](int buf, int st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = w * 4 + i;
            const char* src;
            if (f < 8) {
                int krt = st * 2 + (f >> 2);
                krt = krt < krt_max ? krt : krt_max;
                src = (const char*)(Kb + ((size_t)krt * 4 + (f & 3)) * TILE_ELEMS);
            } else {
                const int g = f - 8, dt = g >> 2;
                int vkt = st * 4 + (g & 3);
                vkt = vkt < vkt_max ? vkt : vkt_max;
                src = (const char*)(Vb + ((size_t)dt * (size_t)(a.Sk_cap >> 4) + (size_t)vkt) * TILE_ELEMS);
            }
            glds16(src + lane * 16, st_base + buf * AT_STAGE_BYTES + f * TILE_BYTES);
        }
    };

    int nst = kst ? kst[0] : Sk_pad / AT_KEYS;
    auto sid = [&](int i) { return kst ? kst[1 + i] : i; };       // i-th visited key stage
    if (MODE == ATT_DEC_SELF) {   // keys beyond the workgroup's last query are all causally masked
        const int last_q = qb * 128 + 127;
        const int lim = last_q / AT_KEYS + 1;
        nst = nst < lim ? nst : lim;
    }

    f32x16 o[2] = {acc_zero(), acc_zero()};
    // row sums of the bf16 P tile come from the matrix pipe too (A = all-ones fragment): every register of `lsum`
    // holds sum_keys P[key][query of this lane]; only register 0 is maintained across rescales
    f32x16 lsum = acc_zero();
    const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
    float m_run = AT_NEG;

    stage(0, sid(0));
    if (MODE == ATT_ENC) load_bidx(sid(0), bnxt);
    __syncthreads();
    for (int sti = 0; sti < nst; ++sti) {
        const int cur = sti & 1;
        const int st = sid(sti);                                      // key stage (64 keys) processed this iteration
        if (sti + 1 < nst) stage(cur ^ 1, sid(sti + 1));
        if (MODE == ATT_ENC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) bcur[i] = bnxt[i];
            if (sti + 1 < nst) load_bidx(sid(sti + 1), bnxt);
        }
        const char* kb = st_base + cur * AT_STAGE_BYTES + lane * 16;
        const char* vb = kb + 8 * TILE_BYTES;
        // S^T tiles: rows = keys, cols = queries
        f32x16 s[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            s[t2] = acc_zero();
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[t2] = mfma32(ld16(kb + (t2 * 4 + kt) * TILE_BYTES), qf[kt], s[t2]);
        }
        // bias + mask + running max
        float mloc = AT_NEG;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = st * AT_KEYS + t2 * 32 + acc_row(r, half);
                float v = s[t2][r];
                bool ok;
                if (MODE == ATT_ENC) {
                    const uint4& wq = bcur[t2 * 2 + (r >> 3)];
                    const uint32_t wd = ((r >> 1) & 3) == 0 ? wq.x : (((r >> 1) & 3) == 1 ? wq.y : (((r >> 1) & 3) == 2 ? wq.z : wq.w));
                    const uint32_t e = (r & 1) ? (wd >> 16) : wd;        // the upper half of the word is masked off below
                    ok = true;
                    // masked key: tab1 entry = AT_NEG, which absorbs the other terms exactly (|score| << ulp(1e30))
                    v += (tab1[(e >> 10) & 63] + thv[(e >> 5) & 31]) + thv[32 + (e & 31)];
                } else if (MODE == ATT_DEC_SELF) {
                    const int dist = qi - key;
                    ok = kmk[key] != 0 && dist >= 0;
                    const int di = dist < 0 ? 0 : (dist < t1n ? dist : t1n - 1);
                    v += tab1[di];
                } else {
                    ok = kmk[key] != 0;
                }
                if (MODE != ATT_ENC) v = ok ? v : AT_NEG;
                s[t2][r] = v;
                mloc = fmaxf(mloc, v);
            }
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = fast_exp(m_run - m_new);
        m_run = m_new;
        // P^T = exp(S^T - m), rounded to bf16; the row sum is taken over the ROUNDED values (ones-MFMA) so that O/l is a
        // convex combination
        uint4 pch[4];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            f32x16 p;
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = fast_exp(s[t2][r] - m_new);
            acc_to_chunks(p, half, &pch[2 * t2]);
        }
        // rescale only when some query of the wave raised its running max (rare after the first stages)
#ifdef MG_EMU
        const bool rescale = true;
#else
        const bool rescale = __any(alpha != 1.0f);
#endif
        if (rescale) {
            lsum[0] *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const uint4 v0 = ld16(vb + (0 * 4 + kk) * TILE_BYTES), v1 = ld16(vb + (1 * 4 + kk) * TILE_BYTES);
            o[0] = mfma32(v0, pch[kk], o[0]);
            o[1] = mfma32(v1, pch[kk], o[1]);
            lsum = mfma32(ones, pch[kk], lsum);
        }
        __syncthreads();
    }

    // O^T / l -> packed context rows [b*Sq_cap + q][h*64 + dim]
    const float inv = 1.0f / lsum[0];
    const int HD = a.H * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        f32x16 v;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = o[dt][r] * inv;
        uint4 ch[2];
        acc_to_chunks(v, half, ch);
        if (qi < a.Sq_cap) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                st16(a.ctx + pk_off(b * a.Sq_cap + qi, h * 64 + dt * 32 + q * 16 + half * 8, HD), ch[q]);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Encoder attention (ATT_ENC), second form.  Workgroup = 8 waves = 256 consecutive queries of one (image, head), 32 per wave;
// one raw barrier per 64-key stage, K / V^T stages double-buffered by inline-asm LDS-DMA that stays in flight across the
// barrier (the builtin form makes hipcc drain vmcnt(0) in front of every ds_read, i.e. it serialised each stage behind the
// prefetch of the next one).  What is new against attention_kernel<ATT_ENC> is the cost of a score (VALU and LDS are the
// co-critical pipes of this kernel next to the matrix pipe: 16 scores per lane and 32x32 tile against 8 MFMAs):
//   * horizontal + vertical bias from ONE lookup: a per-head [vertical bucket][horizontal bucket] table of 1024 sums in LDS
//     (+ one entry = AT_NEG for masked keys); the per-pair index array holds the BYTE address of the entry (u16), so a
//     score costs one and/shift and one ds_read_b32.  Horizontal bucket minor = LDS bank: the 32 queries of a wave that lie
//     in one patch row share the vertical bucket of a key and hit 32 different banks or the same word;
//   * 1-D bias by distance: all pairs of a (32-query, 32-key) tile further apart than 128 positions share one bucket
//     (stock:422-468 saturates at max_distance), so the term is a per-tile constant folded into the running-max
//     bookkeeping (zero instructions per score); only the ~5 tiles around the diagonal read a distance-indexed table, with
//     the register's key offset as the instruction's immediate offset (no address arithmetic per score);
//   * exp2 with log2(e) folded into the score fma and into the tables; row sums of the rounded weights by v_dot2 on the
//     packed pairs (half an instruction per score; the ones-MFMA of the first form cost 20 % of the matrix pipe).
// ---------------------------------------------------------------------------------------------------------
constexpr int AE_WAVES = 8;
constexpr int AE_QB = 32 * AE_WAVES;                 // queries per workgroup
constexpr int AE_HV = 1024;                          // [bv][bh] entries; entry AE_HV = masked key
constexpr int AE_D1 = 192;                           // distance table covers key - query in [-AE_D1, AE_D1]
constexpr int AE_DEPTH = 2;                          // stages in flight ahead of the one being computed
constexpr int AE_RING = AE_DEPTH + 1;                // K / V^T stage buffers (a slot is refilled one barrier after its last read)
constexpr int AE_IDX_BYTES = AE_WAVES * 4 * TILE_BYTES;   // index words of one stage: 4 KiB per wave, read back by that wave only
constexpr float AE_LOG2E = 1.4426950408889634f;
constexpr int AE_MAXST = 64;                         // visited-stage list kept in LDS (S_cap <= 4096)
constexpr int AE_SMEM = AE_RING * AT_STAGE_BYTES + AE_DEPTH * AE_IDX_BYTES + (AE_HV + 4) * 4 + (2 * AE_D1 + 4) * 4 + AE_MAXST * 4;

// Pipeline: the registers of a workgroup (170+ per lane) allow one workgroup per CU, so all memory-level parallelism is the
// prefetch depth: K / V^T stages and the wave's index words all travel by LDS-DMA (inline asm, invisible to hipcc's
// s_waitcnt insertion) AE_DEPTH stages ahead; every wave waits with a COUNTED vmcnt for its own copies of the current stage
// and one raw barrier per stage makes the K / V^T fragments visible to the other waves.  6 copies per wave and stage
// (4 x 1 KiB of index words, 2 fragments; waves without an attended query issue only the 2 fragments).  Inside the stage
// loop there must be NO compiler-tracked vector-memory access: a compiler-inserted vmcnt(0) would drain the pipeline
// (the stage list therefore lives in LDS, and the loads of the prologue are retired by hand before the loop).
// XP != 0: timing experiments of the tools build only (MG_ATT_EXP, WRONG results): 1 = no index-word copies (stale LDS is read),
// 2 = no table lookup (bias 0), 4 = no exp2 (weights = shifted scores), 8 = K / V^T copies of the first two stages only
// QT = 32-query tiles per wave.  QT = 1 (default): 8 waves (two per SIMD), the form of rounds 2-4.  QT = 2 (round 5, measured slower, see the launcher): 4 waves (one per SIMD, up to
// 512 registers per lane), each with TWO query tiles: every K / V^T fragment read from LDS feeds two MFMAs (half the fragment reads
// per query: the LDS port and the dependent read -> MFMA latencies were what a stage waited on, not the matrix pipe) and the two tiles'
// softmax chains are independent instruction streams the scheduler interleaves.  A query's arithmetic is the same sequence of
// operations in both forms: results are bit-identical (tests/test_kernels.py).
// PLAIN (round 5, the ChemicalOCR vision tower: ATT_CROSS over full frames): no relative bias and no key mask - no index words are copied or read, no
// table exists, a score is s * log2(e); the caller guarantees kmask == null and Sk == Sk_cap.  Everything else (stages, softmax bookkeeping, the two
// products) is the encoder's second form.
template <int XP = 0, int QT = 1, bool PLAIN = false>
__global__ __launch_bounds__(512 / QT) void attention_enc_kernel(AttnArgs a) {
    constexpr int NWV = AE_WAVES / QT;                   // waves per workgroup
    constexpr int NTH = 64 * NWV;
    constexpr int FPW = 16 / NWV;                        // K / V^T fragments a wave copies per stage
    MG_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef MG_EMU
    const int w = tid >> 6;
#else
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: tile distances below are scalar branches
#endif
    const int half = lane >> 5, l32 = lane & 31;
    const int nqb = (a.Sq_cap + AE_QB - 1) / AE_QB;
    int bid = blockIdx.x;
    if ((a.B & 7) == 0) {          // an image's heads and query blocks on one XCD: its index array and K/V reach one L2
        const int per_img = a.H * nqb, xcd = bid & 7, j = bid >> 3;
        bid = ((j / per_img) * 8 + xcd) * per_img + (j % per_img);
    }
    const int qb = bid % nqb;
    const int bh = bid / nqb;
    const int h = bh % a.H, b = bh / a.H;
    const int HD = a.H * 64;
    const int* kst = a.kst ? a.kst + (size_t)b * (1 + (a.Sk_cap >> 6)) : nullptr;
    if (a.qbv) {                   // query block without an attended position: clear its context rows, nothing else
        const int n128 = (a.Sq_cap + 127) / 128;
        const uint8_t* qv = a.qbv + (size_t)b * n128;
        const bool any = qv[2 * qb] | ((2 * qb + 1 < n128) ? qv[2 * qb + 1] : 0);
        if (!any) {
            for (int i = tid; i < AE_QB * 8; i += NTH) {
                const int q = qb * AE_QB + (i >> 3), c = (i & 7) * 8;
                if (q < a.Sq_cap) st16(a.ctx + pk_off(b * a.Sq_cap + q, h * 64 + c, HD), make_uint4(0, 0, 0, 0));
            }
            return;
        }
    }
    char* st_base = smem;
    char* ix_base = smem + AE_RING * AT_STAGE_BYTES + w * (4 * QT * TILE_BYTES);      // + slot * AE_IDX_BYTES; tile t of the wave: + t * 4 KiB
    float* hv = (float*)(smem + AE_RING * AT_STAGE_BYTES + AE_DEPTH * AE_IDX_BYTES);
    float* t1d = hv + AE_HV + 4;
    int* ksl = (int*)(t1d + 2 * AE_D1 + 4);
    const int nst = kst ? kst[0] : (a.Sk + AT_KEYS - 1) / AT_KEYS;
    for (int i = tid; i < nst && i < AE_MAXST; i += NTH) ksl[i] = kst ? kst[1 + i] : i;
    if constexpr (!PLAIN) {
        for (int i = tid; i < AE_HV; i += NTH)
            hv[i] = (a.tabh[(size_t)(i & 31) * a.H + h] + a.tabv[(size_t)(i >> 5) * a.H + h]) * AE_LOG2E;
        if (tid < 4) hv[AE_HV + tid] = AT_NEG;
        for (int i = tid; i < 2 * AE_D1 + 1; i += NTH) {
            int d = i - AE_D1;
            d = d < -128 ? -128 : (d > 128 ? 128 : d);
            t1d[i] = a.tab1[(size_t)a.bk1[d + 128] * a.H + h] * AE_LOG2E;
        }
    }

    const int q0w = qb * AE_QB + w * 32 * QT;            // first query of the wave; tile t starts at q0w + 32 t
    const int qrt_max = (a.Sq_cap >> 5) - 1;
    mg_raw16 qr[QT][4];
    int qi[QT];
    const uint16_t* bix[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        int qrt = (q0w + 32 * t) >> 5;
        if (qrt > qrt_max) qrt = qrt_max;
        const uint16_t* Qb = a.Q + (((size_t)b * a.H + h) * (size_t)(a.Sq_cap >> 5) + (size_t)qrt) * (4 * TILE_ELEMS);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) gld16_async(qr[t][kt], (const char*)(Qb + kt * TILE_ELEMS) + lane * 16);
        qi[t] = q0w + 32 * t + l32;
        const int qcl = qi[t] < a.Sk_cap ? qi[t] : a.Sk_cap - 1;
        bix[t] = PLAIN ? nullptr : a.bidx + ((size_t)b * (size_t)(a.Sk_cap >> 5) * (size_t)a.Sk_cap + (size_t)qcl) * 32 + half * 16;
    }
    // a wave in a 128-query block without an attended position (the granularity of attn_lists, and what the first form of
    // the kernel skipped: padded rows next to attended ones are still computed, as the reference does) keeps loading its
    // share of the stages and meeting the barriers, but computes nothing; its context rows are cleared (later GEMMs must
    // see finite values).  (QT = 2: the wave's 64 queries lie inside one 128-query block.)
    int act = q0w < a.Sq_cap ? 1 : 0;
    if (a.qbv && act) act = a.qbv[(size_t)b * ((a.Sq_cap + 127) / 128) + (q0w >> 7)] ? 1 : 0;
    __syncthreads();               // tables and stage list complete
    // retire every load of the prologue by hand (the Q fragments were raw loads: nothing may touch them before this wait):
    // from here on the vector-memory queue holds only the hand-counted copies below
    MG_WAIT_VMCNT_TIE4(0, qr[0][0], qr[0][1], qr[0][2], qr[0][3]);
    if constexpr (QT == 2) { MG_TIE(qr[QT - 1][0]); MG_TIE(qr[QT - 1][1]); MG_TIE(qr[QT - 1][2]); MG_TIE(qr[QT - 1][3]); }
    uint4 qf[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) qf[t][kt] = raw16_get(qr[t][kt]);
    // 1-D term of a tile whose pairs are all >= 128 positions apart (saturated bucket, stock:455-466): keys left / right
    const float c_left = PLAIN ? 0.f : t1d[0], c_right = PLAIN ? 0.f : t1d[2 * AE_D1];

    const size_t bix_tile = (size_t)a.Sk_cap * 32;
    const uint16_t* Kb = a.K + ((size_t)b * a.H + h) * (size_t)(a.Sk_cap >> 5) * (4 * TILE_ELEMS);
    const uint16_t* Vb = a.Vt + ((size_t)b * a.H + h) * 2 * (size_t)(a.Sk_cap >> 4) * TILE_ELEMS;
    const int krt_max = (a.Sk_cap >> 5) - 1, vkt_max = (a.Sk_cap >> 4) - 1;
    auto sid = [&](int i) { return ksl[i]; };
    // issue everything this wave copies for the i-th visited stage: its index words (active waves; 16 keys = 32 B per lane
    // and 32-key tile, as 4 x 16 B per lane -> 4 copies of 1 KiB per query tile) into index slot i % AE_DEPTH, then its K / V^T
    // fragments into ring slot i % AE_RING (fragments 0..7 = K: key-tile f/4, dk-tile f%4; 8..15 = V^T: d-tile, key-k-tile)
    auto issue = [&](int i) {
        const int st = sid(i);
        if (act && !(XP & 1) && !PLAIN) {
            char* ix = ix_base + (i % AE_DEPTH) * AE_IDX_BYTES;
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const uint16_t* p = bix[t] + (size_t)(st * 2 + t2) * bix_tile;
                    glds16_async(p, ix + (t * 4 + t2 * 2) * TILE_BYTES);
                    glds16_async(p + 8, ix + (t * 4 + t2 * 2 + 1) * TILE_BYTES);
                }
        }
        char* dst = st_base + (i % AE_RING) * AT_STAGE_BYTES;
#pragma unroll
        for (int k = 0; k < FPW; ++k) {
            const int f = w * FPW + k;
            const char* src;
            if (f < 8) {
                int krt = st * 2 + (f >> 2);
                krt = krt < krt_max ? krt : krt_max;
                src = (const char*)(Kb + ((size_t)krt * 4 + (f & 3)) * TILE_ELEMS);
            } else {
                const int g = f - 8, dt = g >> 2;
                int vkt = st * 4 + (g & 3);
                vkt = vkt < vkt_max ? vkt : vkt_max;
                src = (const char*)(Vb + ((size_t)dt * (size_t)(a.Sk_cap >> 4) + (size_t)vkt) * TILE_ELEMS);
            }
            if (!(XP & 8) || i < 2) glds16_async(src + lane * 16, dst + f * TILE_BYTES);
        }
    };
    constexpr int XW_ACT = ((XP & 1) ? 0 : 4) + ((XP & 8) ? 0 : 2), XW_IDLE = (XP & 8) ? 0 : 2;      // copies per wave and stage (QT = 1 experiments)

    f32x16 o[QT][2];
    float lsum[QT], m_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) { o[t][0] = acc_zero(); o[t][1] = acc_zero(); lsum[t] = 0.f; m_run[t] = AT_NEG; }
    static_assert(AE_DEPTH == 2, "wait counts below");
    issue(0);
    if (1 < nst) issue(1);
    for (int sti = 0; sti < nst; ++sti) {
        // own copies of stage sti have landed when at most those of the one later stage in flight are outstanding
        if constexpr (XP == 0 && PLAIN) {        // (two fragment copies per wave and stage, active or not)
            static_assert(QT == 1 || !PLAIN, "the plain form exists for one query tile per wave");
            if (sti + 1 < nst && !(a.dbg & 1)) MG_WAIT_VMCNT(2); else MG_WAIT_VMCNT(0);
        } else if constexpr (XP == 0) {
            if (sti + 1 < nst && !(a.dbg & 1)) {
                if constexpr (QT == 1) { if (act) MG_WAIT_VMCNT(6); else MG_WAIT_VMCNT(2); }
                else { if (act) MG_WAIT_VMCNT(12); else MG_WAIT_VMCNT(4); }
            } else MG_WAIT_VMCNT(0);
        } else {                       // (experiments: same structure with the variant's copy counts)
            if (sti + 1 < nst && sti >= 2) {
                if (act) { if constexpr (XW_ACT == 6) MG_WAIT_VMCNT(6); else if constexpr (XW_ACT == 4) MG_WAIT_VMCNT(4); else if constexpr (XW_ACT == 2) MG_WAIT_VMCNT(2); else MG_WAIT_VMCNT(0); }
                else { if constexpr (XW_IDLE == 2) MG_WAIT_VMCNT(2); else MG_WAIT_VMCNT(0); }
            } else MG_WAIT_VMCNT(0);
        }
        MG_BARRIER_RAW();            // everybody's copies of stage sti have landed; the K / V^T slot of stage sti - 1 is free
        const int st = sid(sti);
        if (!act) {
            if (sti + AE_DEPTH < nst) issue(sti + AE_DEPTH);
            continue;
        }
        uint4 bcur[QT][4];
        if constexpr (!PLAIN) {
            const char* ix = ix_base + (sti % AE_DEPTH) * AE_IDX_BYTES + lane * 16;
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) bcur[t][i] = ld16(ix + (t * 4 + i) * TILE_BYTES);
            MG_WAIT_LGKM0();         // index words are in registers: their slot is refilled for stage sti + AE_DEPTH
        }
        if (sti + AE_DEPTH < nst) issue(sti + AE_DEPTH);
        const char* kb = st_base + (sti % AE_RING) * AT_STAGE_BYTES + lane * 16;
        const char* vb = kb + 8 * TILE_BYTES;
        f32x16 s[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) { s[t][0] = acc_zero(); s[t][1] = acc_zero(); }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {           // the two key tiles alternate: independent accumulator chains; a fragment feeds QT MFMAs
            const uint4 k0f = ld16(kb + (0 * 4 + kt) * TILE_BYTES), k1f = ld16(kb + (1 * 4 + kt) * TILE_BYTES);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                s[t][0] = mfma32(k0f, qf[t][kt], s[t][0]);
                s[t][1] = mfma32(k1f, qf[t][kt], s[t][1]);
            }
        }
        uint4 pch[QT][4];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const int q0 = q0w + 32 * t;
            // scores in the log2 domain: v = s*log2e + hv[pair] (+ 1-D term near the diagonal); tile maxima
            float cst[2], tmx[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int k0 = st * AT_KEYS + t2 * 32;
                const int dk = k0 - q0;                                   // wave-uniform
                const uint32_t* bw = (const uint32_t*)&bcur[t][t2 * 2];
                float tm = AT_NEG;
                if constexpr (PLAIN) {
                    (void)bw; (void)dk;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = s[t][t2][r] * AE_LOG2E;
                        s[t][t2][r] = v;
                        tm = fmaxf(tm, v);
                    }
                    cst[t2] = 0.f;
                } else if (dk > -(128 + 31) && dk < 128 + 31) {                 // some pair of the tile is closer than 128: per-score term
                    const char* tl = (const char*)t1d + (k0 - qi[t] + 4 * half + AE_D1) * 4;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t e = (r & 1) ? (bw[r >> 1] >> 16) : (bw[r >> 1] & 0xFFFFu);
                        const float bias = ((XP & 2) ? __uint_as_float(e) : *(const float*)((const char*)hv + e)) + *(const float*)(tl + ((r & 3) + 8 * (r >> 2)) * 4);
                        const float v = fmaf(s[t][t2][r], AE_LOG2E, bias);
                        s[t][t2][r] = v;
                        tm = fmaxf(tm, v);
                    }
                    cst[t2] = 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t e = (r & 1) ? (bw[r >> 1] >> 16) : (bw[r >> 1] & 0xFFFFu);
                        const float v = fmaf(s[t][t2][r], AE_LOG2E, (XP & 2) ? __uint_as_float(e) : *(const float*)((const char*)hv + e));
                        s[t][t2][r] = v;
                        tm = fmaxf(tm, v);
                    }
                    cst[t2] = dk < 0 ? c_left : c_right;
                }
                tmx[t2] = tm + cst[t2];
            }
            float mloc = fmaxf(tmx[0], tmx[1]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            const float m_new = fmaxf(m_run[t], mloc);
            const float alpha = fast_exp2(m_run[t] - m_new);
            m_run[t] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const float mt = m_new - cst[t2];
                f32x16 p;
#ifdef MG_EMU
#pragma unroll
                for (int r = 0; r < 16; ++r) p[r] = fast_exp2(s[t][t2][r] - mt);
#else
                typedef float mg_f32x2 __attribute__((ext_vector_type(2)));
                const mg_f32x2 mt2 = {mt, mt};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {                      // packed subtract: half an instruction per score
                    const mg_f32x2 sv = {s[t][t2][r], s[t][t2][r + 1]};
                    const mg_f32x2 d = sv - mt2;
                    if constexpr (XP & 4) { p[r] = d.x; p[r + 1] = d.y; } else { p[r] = fast_exp2(d.x); p[r + 1] = fast_exp2(d.y); }
                }
#endif
                const PackedAcc pa = acc_pack(p);
                packed_to_chunks_swap(pa, half, &pch[t][2 * t2]);
            }
            // the row sum is taken over the ROUNDED values (two per v_dot2 with a ones pair), so that O / l stays a convex
            // combination of the value rows whatever the rounding of the dominant weights; summed after the half-wave exchange
            // (which only permutes a query's weights between its two lanes), so the packed words die right there
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                psum = dot2_bf16(pch[t][c].x, 0x3F803F80u, psum); psum = dot2_bf16(pch[t][c].y, 0x3F803F80u, psum);
                psum = dot2_bf16(pch[t][c].z, 0x3F803F80u, psum); psum = dot2_bf16(pch[t][c].w, 0x3F803F80u, psum);
            }
#ifdef MG_EMU
            const bool rescale = true;
#else
            const bool rescale = __any(alpha != 1.0f);
#endif
            if (rescale) {
                lsum[t] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][dt][r] *= alpha;
            }
            lsum[t] += psum;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const uint4 v0 = ld16(vb + (0 * 4 + kk) * TILE_BYTES), v1 = ld16(vb + (1 * 4 + kk) * TILE_BYTES);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                o[t][0] = mfma32(v0, pch[t][kk], o[t][0]);
                o[t][1] = mfma32(v1, pch[t][kk], o[t][1]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float ls = lsum[t];
        ls += __shfl_xor(ls, 32);          // the two halves hold different keys of the same query
        const float inv = act ? 1.0f / ls : 0.f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            f32x16 v;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = o[t][dt][r] * inv;
            uint4 ch[2];
            acc_to_chunks(v, half, ch);
            if (qi[t] < a.Sq_cap) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    st16(a.ctx + pk_off(b * a.Sq_cap + qi[t], h * 64 + dt * 32 + q * 16 + half * 8, HD), ch[q]);
            }
        }
    }
}

// Measured and rejected in round 4 (kernels in this file's history, numbers under profiles/r04_e_*): (1) the same arithmetic on a
// PING-PONG schedule - a stage cut into an M phase (P.V of stage j-1, K.Q^T of stage j; fragments prefetched into registers, index
// words straight from global memory into registers) and a V phase (the softmax arithmetic), the two wave groups one barrier apart:
// bit-identical, encoder 37.1 against 36.5 ms; phase stamps show ~2000 cycles per stage in the V phase against ~900 in the M phase;
// (2) 12 waves = 384 queries per workgroup (three waves per SIMD at 146 registers): bit-identical, 37.5 ms.  Neither more waves nor
// complementary phases move it: the stage is a chain of dependent LDS / matrix / transcendental latencies, not a busy pipe.

static size_t attn_smem(const AttnArgs& a) {
    const int Sk_pad = (a.Sk + AT_KEYS - 1) / AT_KEYS * AT_KEYS;
    const int t1n = (a.mode == ATT_CROSS) ? 0 : (a.mode == ATT_ENC ? 64 : a.tab1_len);
    size_t sz = 2 * AT_STAGE_BYTES + (size_t)((t1n + 3) & ~3) * 4;
    if (a.mode == ATT_ENC) sz += 64 * 4;
    else sz += (size_t)Sk_pad + 16;
    return (sz + 15) & ~(size_t)15;
}

// Per-(image, query, key) bias bucket indices of the encoder, once per batch (the three relative biases are shared
// by all layers and differ between heads only through the table values, stock:1215,1234-1235):
//   entry = BYTE address of the pair's sum in the per-head [vertical bucket][horizontal bucket] table of
//   attention_enc_kernel: 4 * (32 * bucketV + bucketH), or 4 * 1024 (the masked entry) for a key that is not attended
//   (stock:904-1009 semantics: box centres in float64, difference -> fp32, *100, truncation).  The 1-D bucket is a function
//   of key - query in the combined sequence and is not stored.
// Layout [image][key tile of 32][query][32] with the 32 keys of a tile in MFMA accumulator order
// (entry half*16 + r <-> key (r%4) + 8*(r/4) + 4*half), so a lane of the attention kernel reads its 16 keys as 32 B.
__global__ __launch_bounds__(256) void bias_index_kernel(uint16_t* out, const double* cx, const double* cy, const uint8_t* kmask,
                                                    const int* bk1, const int* bkhv, int B, int Sk, int S_cap) {
    const size_t total = (size_t)B * (size_t)(S_cap >> 5) * (size_t)S_cap * 32;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 31);
        size_t rest = i >> 5;
        const int q = (int)(rest % (size_t)S_cap);
        rest /= (size_t)S_cap;
        const int kt = (int)(rest % (size_t)(S_cap >> 5)), b = (int)(rest / (size_t)(S_cap >> 5));
        const int r = e & 15, half = e >> 4;
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        uint32_t v = 4u * AE_HV;
        if (key < Sk && (kmask == nullptr || kmask[(size_t)b * S_cap + key] != 0)) {
            const double qx = cx[(size_t)b * S_cap + q], qy = cy[(size_t)b * S_cap + q];
            const float fx = (float)(cx[(size_t)b * S_cap + key] - qx) * 100.0f;
            const float fy = (float)(cy[(size_t)b * S_cap + key] - qy) * 100.0f;
            const int dx = (int)fmaxf(fminf(fx, 100.0f), -100.0f);
            const int dy = (int)fmaxf(fminf(fy, 100.0f), -100.0f);
            v = 4u * (32u * (uint32_t)bkhv[dy + 100] + (uint32_t)bkhv[dx + 100]);
        }
        out[i] = (uint16_t)v;
    }
}
void bias_index(uint16_t* out, const double* cx, const double* cy, const uint8_t* kmask, const int* bk1, const int* bkhv, int B,
                int Sk, int S_cap, mgStream_t stream) {
    MG_LAUNCH(bias_index_kernel, dim3(4096), dim3(256), 0, stream, out, cx, cy, kmask, bk1, bkhv, B, Sk, S_cap);
}

// one workgroup per image: which 64-key stages / 128-query blocks contain an attended position
__global__ __launch_bounds__(64) void attn_lists_kernel(const uint8_t* kmask, int Sk, int S_cap, int* kst, uint8_t* qbv) {
    MG_DYN_SMEM(smem);
    unsigned char* flag = (unsigned char*)smem;              // [S_cap/64]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nstg = S_cap >> 6, nqb = (S_cap + 127) / 128;
    const uint8_t* mk = kmask + (size_t)b * S_cap;
    for (int s = tid; s < nstg; s += 64) {
        int any = 0;
        for (int k = s * 64; k < s * 64 + 64 && k < Sk; ++k) any |= mk[k];
        flag[s] = any ? 1 : 0;
    }
    __syncthreads();
    for (int q = tid; q < nqb; q += 64) qbv[(size_t)b * nqb + q] = (flag[2 * q] | ((2 * q + 1 < nstg) ? flag[2 * q + 1] : 0)) ? 1 : 0;
    if (tid == 0) {
        int* out = kst + (size_t)b * (1 + nstg);
        int n = 0;
        for (int s = 0; s < nstg; ++s)
            if (flag[s]) out[1 + n++] = s;
        if (n == 0) { out[1] = 0; n = 1; }                   // nothing attended: keep one stage so the softmax stays finite
        out[0] = n;
    }
}
void attn_lists(const uint8_t* kmask, int B, int Sk, int S_cap, int* kst, uint8_t* qbv, mgStream_t stream) {
    MG_LAUNCH(attn_lists_kernel, dim3(B), dim3(64), (size_t)((S_cap >> 6) + 16), stream, kmask, Sk, S_cap, kst, qbv);
}

// Ascending list of the 32-row tiles of the [B][S_cap] row space that hold at least one attended position, and its length
// (GemmArgs::row_tiles).  One workgroup: flags in parallel, the compaction by one thread in tile order (deterministic; a few
// thousand tiles at most).
__global__ __launch_bounds__(256) void row_tile_list_kernel(const uint8_t* kmask, int n_tiles, int* list, int* count) {
    MG_DYN_SMEM(smem);
    int* base = (int*)smem;                                  // [256] live tiles in front of each thread's chunk
    const int tid = threadIdx.x;
    const int per = (n_tiles + 255) / 256, t0 = tid * per, t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    int mine = 0;
    for (int t = t0; t < t1; ++t) {
        const uint4* p = (const uint4*)(kmask + (size_t)t * 32);
        const uint4 a = p[0], b = p[1];
        mine += (a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) ? 1 : 0;
    }
    base[tid] = mine;
    __syncthreads();
    if (tid == 0) {                                          // exclusive scan over the 256 chunk counts (ascending tile order)
        int run = 0;
        for (int i = 0; i < 256; ++i) { const int c = base[i]; base[i] = run; run += c; }
        if (run == 0) { list[0] = 0; run = 1; }              // nothing attended anywhere: keep one tile (the GEMMs need M >= 1)
        *count = run;
    }
    __syncthreads();
    int n = base[tid];
    for (int t = t0; t < t1; ++t) {
        const uint4* p = (const uint4*)(kmask + (size_t)t * 32);
        const uint4 a = p[0], b = p[1];
        if (a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) list[n++] = t;
    }
}
void row_tile_list(const uint8_t* kmask, int rows, int* list, int* count, mgStream_t stream) {
    const int n_tiles = rows >> 5;                               // rows is a multiple of 32
    MG_LAUNCH(row_tile_list_kernel, dim3(1), dim3(256), (size_t)(256 * sizeof(int)), stream, kmask, n_tiles, list, count);
}

static std::atomic<int> g_att_qt{-1};          // query tiles per wave of the encoder attention: -1 = MG_ATT_QT or the default (2); process-wide test / A-B switch
void attention_set_qt(int qt) { g_att_qt = qt; }

void attention(const AttnArgs& a_in, mgStream_t stream) {
    AttnArgs a = a_in;
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("MG_ATT_DBG"); dbg = e ? atoi(e) : 0; }
    a.dbg = dbg;
    const int nqb = (a.Sq_cap + 127) / 128;
    const dim3 grid(a.B * a.H * nqb), block(256);
    const size_t sh = attn_smem(a);
    if (a.mode == ATT_ENC) {
        const int nqe = (a.Sq_cap + AE_QB - 1) / AE_QB;
        static bool once = false;
        if (!once) { MG_SET_MAX_SMEM((&attention_enc_kernel<0, 1>), AE_SMEM); once = true; }
#ifdef MG_TOOLS      // what-if variants with WRONG results: tools builds only
        static int xp = -1;
        if (xp < 0) { const char* e = getenv("MG_ATT_EXP"); xp = e ? atoi(e) : 0; }
        if (xp) {
#define MG_AX(N) case N: { static bool o = false; if (!o) { MG_SET_MAX_SMEM(&attention_enc_kernel<N>, AE_SMEM); o = true; } \
                           MG_LAUNCH(attention_enc_kernel<N>, dim3(a.B * a.H * nqe), dim3(512), (size_t)AE_SMEM, stream, a); } break;
            switch (xp) { MG_AX(1) MG_AX(2) MG_AX(3) MG_AX(4) MG_AX(7) MG_AX(8) MG_AX(9) MG_AX(15) default: MG_AX(11) }
#undef MG_AX
            return;
        }
#endif
        // Measured and rejected in round 5 (profiles/r05_g_attention_two_tiles_per_wave.txt): QT = 2 - two query tiles per wave, 4 waves per
        // workgroup (one per SIMD, 168 + accumulator registers), every K / V^T fragment read feeding two MFMAs: bit-identical, 561 us per
        // launch against 430 us for the 8-wave form at the benchmark shape (encoder 36.4 against 33.2 ms per batch on the same box): two
        // WAVES per SIMD hide the stage's dependent LDS -> MFMA -> exp2 latencies, two instruction streams inside one wave do not (the
        // near-diagonal branches are per tile and keep the scheduler from interleaving them).  Kept behind MG_ATT_QT=2 / mgk_set_attention_qt.
        int qt = g_att_qt;
        if (qt < 0) { const char* e = getenv("MG_ATT_QT"); qt = e ? atoi(e) : 1; g_att_qt = qt; }
        if (qt == 2) {
            static bool once2 = false;
            if (!once2) { MG_SET_MAX_SMEM((&attention_enc_kernel<0, 2>), AE_SMEM); once2 = true; }
            MG_LAUNCH((attention_enc_kernel<0, 2>), dim3(a.B * a.H * nqe), dim3(256), (size_t)AE_SMEM, stream, a);
        } else {
            MG_LAUNCH((attention_enc_kernel<0, 1>), dim3(a.B * a.H * nqe), dim3(512), (size_t)AE_SMEM, stream, a);
        }
    } else if (a.mode == ATT_DEC_SELF) MG_LAUNCH((attention_kernel<ATT_DEC_SELF>), grid, block, sh, stream, a);
    else {
        // bias-free attention over full rows of keys (the ChemicalOCR vision tower: 1024 patches per frame, no mask unless a frame is padded): the
        // encoder's second-form kernel without its bias path - 253 -> us per launch at 32 frames (profiles/r05_s_*); MG_ATT_PLAIN=0: the first form
        static int plain = -1;
        if (plain < 0) { const char* e = getenv("MG_ATT_PLAIN"); plain = e ? atoi(e) : 1; }
        if (plain && !a.kmask && a.Sk == a.Sk_cap && a.Sq == a.Sq_cap && (a.Sk_cap & 63) == 0 && (a.Sq_cap % AE_QB) == 0 && (a.Sk_cap >> 6) <= AE_MAXST && !a.kst && !a.qbv) {
            const int nqe = a.Sq_cap / AE_QB;
            static bool oncep = false;
            if (!oncep) { MG_SET_MAX_SMEM((&attention_enc_kernel<0, 1, true>), AE_SMEM); oncep = true; }
            MG_LAUNCH((attention_enc_kernel<0, 1, true>), dim3(a.B * a.H * nqe), dim3(512), (size_t)AE_SMEM, stream, a);
            return;
        }
        MG_LAUNCH((attention_kernel<ATT_CROSS>), grid, block, sh, stream, a);
    }
}

}  // namespace mg
