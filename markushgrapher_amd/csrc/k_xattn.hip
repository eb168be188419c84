// Weight-absorbed cross-attention of the greedy decode step (stock modeling_udop.py:524-575, the EncDecAttention of a decoder block).
//
// Stock computes per decoder layer K_l = enc·Wk_l^T and V_l = enc·Wv_l^T once per image and every step reads both (2 x 2·d_kv·H bytes
// per encoder position and layer: the dominant HBM stream of decoding).  The encoder states `enc` are the same for every layer, and
//     softmax(q_h · K_h^T) · V_h  =  [ softmax((q_h · Wk_h) · enc^T) · enc ] · Wv_h^T            (per head h; cross-attention has no bias term)
// so a layer needs only the states themselves: one stream of 2·d bytes per position, used twice from LDS - scores
// S[head][key] = q'_h · enc[key] with q'_h = q_h·Wk_h (d features), and context c_h = sum_key P[head][key] · enc[key] (d features)
// which a small per-head projection maps back to d_kv.  Bytes per position and layer: 4·H·d_kv -> 2·d (half, for d = H·d_kv); the
// cross-K/V projections of the encoder phase (12 % of its flops) and the per-layer K/V buffers disappear.
//
//   xq_expand_kernel      q [rows][H][64]  ->  q' [rows][H][d]          (per head a [rows x 64] x [64 x d] product, MFMA 16x16x32)
//   xattn_stream_kernel   q', enc          ->  c [rows][split][H][d] bf16, normalised per split, + (m, l) per head   (the HBM stream; MFMA)
//   xctx_contract_kernel  c, (m, l)        ->  ctx [rows][H*64] bf16     (merge of the key splits, 1 / l, per head [rows x d] x [d x 64])
//
// Rounding points: q (bf16, as before), q' (bf16), P (bf16), c (fp32 accumulation -> bf16 after the normalisation), ctx (bf16, as before).
#include "mg_kernels.h"

namespace mg {

constexpr float XA_NEG = -1.0e30f;
constexpr int XA_KEYS = 16;          // keys per stage of the stream
constexpr float XA_DEFER = 8.0f;     // the running maximum may lag by this much (log2 units) before the accumulators are rescaled

// LDS image of a stage: [16 keys][d/8 chunks of 16 B], chunk j of key k stored at chunk position j ^ xa_swz(k) (low bits only), so that
// the 16 lanes of a ds_read_b128 group (16 different keys, two neighbouring chunks) and the 32 lanes of a transposed read (8 keys x 2
// chunks x 2 halves) each cover all 64 banks once.
MG_HD int xa_swz(int k) {
    const int hi = k >> 3, kk = (k & 7) ^ (hi ? 4 : 0);
    return ((kk << 1) | hi) & 15;
}

// ---- q' = q_h · Wk_h ----------------------------------------------------------------------------------------------------
// grid (H, d / (32 * waves)); wave: 32 features (two 16-row MFMA tiles whose rows are interleaved 4 by 4, so that a lane ends with 8
// consecutive features of one sequence row = one 16-byte store).  wk: [H][d][64] bf16 (Wk_h transposed: feature-major).
__global__ __launch_bounds__(256) void xq_expand_kernel(XAttnArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const int h = blockIdx.x, f0 = (blockIdx.y * nw + w) * 32;
    const int m = lane & 15, g = lane >> 4;
    const uint16_t* wrow = a.wk + ((size_t)h * a.d + f0 + 8 * (m >> 2) + (m & 3)) * 64 + 8 * g;
    uint4 wa[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wa[t][ks] = ld16(wrow + (size_t)t * 4 * 64 + ks * 32);
    for (int r0 = 0; r0 < a.rows; r0 += 16) {
        int row = r0 + m;
        const bool in = row < a.rows;
        row = in ? row : a.rows - 1;
        const uint16_t* qp = a.q + ((size_t)row * a.H + h) * 64 + 8 * g;
        const uint4 q0 = ld16(qp), q1 = ld16(qp + 32);
        f32x4 c0 = acc4_zero(), c1 = acc4_zero();
        c0 = mfma16(wa[0][0], q0, c0); c0 = mfma16(wa[0][1], q1, c0);
        c1 = mfma16(wa[1][0], q0, c1); c1 = mfma16(wa[1][1], q1, c1);
        if (in)
            st16(a.qx + ((size_t)row * a.H + h) * a.d + f0 + 8 * g,
                 make_uint4(pack_bf16(c0[0], c0[1]), pack_bf16(c0[2], c0[3]), pack_bf16(c1[0], c1[1]), pack_bf16(c1[2], c1[3])));
    }
}

// ---- the stream -------------------------------------------------------------------------------------------------------------
// One workgroup = (sequence row, key split).  NG wave groups of NW = d / (16 NF) waves each: group gi takes the stages gi, gi + NG, ...
// of the split (a stage = 16 keys), wave wq of a group owns features [16 NF wq, 16 NF (wq + 1)).  Two groups = two waves per SIMD whose
// dependency chains (LDS round trips, MFMA results, the softmax's lane exchanges) overlap; each group keeps its own online-softmax state
// and accumulators, the groups are merged once at the end through LDS in group order (a fixed function of the row's keys: a row's bits do
// not depend on what else the call holds).
// Per stage (32·d bytes, copied global -> LDS by DMA into a ring of `nstg` stages, by the waves of the group that consumes it):
//   scores   S^T[key][head] partial over the wave's features: A = enc rows (ds_read_b128), B = q'^T (registers), mfma 16x16x32;
//            partials of the group's NW waves summed through LDS in wave order (every wave ends with the complete tile);
//   softmax  online, per head (= lane % 16): a lane holds 4 keys of one head, the 4 lane groups the other 12;
//   context  c^T[feature][head] += enc^T[feature][key] · P^T[key][head]: A = transposed LDS read of the same stage (4 keys x 16 features
//            per 16 lanes), B = the rounded weights exactly as the score tile left them (k order of mfma 16x16x16 = row order of the
//            16x16x32 result), accumulators [NF][4] per lane.
template <int KS>
MG_DEV void xa_wait_copies(int later) {        // at most `later` stages' copies of this wave (KS each) may remain outstanding
    if (later <= 0) MG_WAIT_VMCNT(0);
    else if (later == 1) { if constexpr (KS == 1) MG_WAIT_VMCNT(1); else if constexpr (KS == 2) MG_WAIT_VMCNT(2); else if constexpr (KS == 4) MG_WAIT_VMCNT(4); else if constexpr (KS == 6) MG_WAIT_VMCNT(6); else MG_WAIT_VMCNT(8); }
    else if (later == 2) { if constexpr (KS == 1) MG_WAIT_VMCNT(2); else if constexpr (KS == 2) MG_WAIT_VMCNT(4); else if constexpr (KS == 4) MG_WAIT_VMCNT(8); else if constexpr (KS == 6) MG_WAIT_VMCNT(12); else MG_WAIT_VMCNT(16); }
    else { if constexpr (KS == 1) MG_WAIT_VMCNT(3); else if constexpr (KS == 2) MG_WAIT_VMCNT(6); else if constexpr (KS == 4) MG_WAIT_VMCNT(12); else if constexpr (KS == 6) MG_WAIT_VMCNT(18); else MG_WAIT_VMCNT(24); }
}

template <int NF, int NW, int NG>
__global__ __launch_bounds__(NW * NG * 64) void xattn_stream_kernel(XAttnArgs a) {
    MG_DYN_SMEM(smem);
    constexpr int KS = NF / 2;                       // 32-feature k-steps of the score product per wave
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef MG_EMU
    const int w = tid >> 6;
#else
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform for the compiler: scalar copy addresses and branches)
#endif
    const int gi = w / NW, wq = w - gi * NW;         // wave group, feature slice inside it
    const int row = blockIdx.x / a.nsplit, split = blockIdx.x - row * a.nsplit;
    if (a.live && a.live[row] == 0) return;          // finished / idle row (whole workgroup)
    const int owner = a.kv_owner ? a.kv_owner[row] : row;
    const int nkeys = a.len[owner];
    const int d = a.d, H = a.H, nch = d >> 3;        // 16-byte chunks per key
    const int head = lane & 15, g = lane >> 4;
    const int fb = wq * 16 * NF;
    const int stage_bytes = XA_KEYS * d * 2;
    const int R = a.nstg / NG;                       // ring depth in iterations (an iteration = one stage per group)
    char* ring = smem;
    float* red = (float*)(smem + (size_t)R * NG * stage_bytes) + (size_t)gi * NW * 256;       // [group][NW][64][4]
    // stages of this split, and of this group
    const int nst_all = (nkeys + XA_KEYS - 1) / XA_KEYS;
    const int per = (nst_all + a.nsplit - 1) / a.nsplit;
    const int st0 = split * per, st1 = (st0 + per < nst_all) ? st0 + per : nst_all;
    const int nst = st1 > st0 ? st1 - st0 : 0;
    const int nit = (nst + NG - 1) / NG;             // iterations of the workgroup
    const int nst_g = nst > gi ? (nst - gi + NG - 1) / NG : 0;      // stages of this group

    // q'^T fragments of the wave's features: lane (head, g) holds q'[head][fb + 32 ks + 8 g .. + 8]
    // (raw loads: the compiler must not know them, or it guards their first use inside the loop with an s_waitcnt vmcnt(0) that
    //  drains the copies in flight every stage)
    mg_raw16 qr[KS];
    {
        const uint16_t* qp = a.qx + ((size_t)row * H + (head < H ? head : 0)) * d + fb + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) gld16_async(qr[ks], qp + ks * 32);
    }
    // deferred RMSNorm of the query row: the scale is applied to the scores (linear in q); fixed summation order
    float qs = 1.0f;
    if (a.qrs.part) {
        float t = 0.f;
        for (int i = lane; i < a.qrs.nparts; i += 64) t += a.qrs.part[(size_t)row * a.qrs.nparts + i];
        t = sum_slots(sum8(t), lane);
        qs = rsqrtf(t * a.qrs.inv_d + a.qrs.eps);
    }
    // Copies of a stage: d/32 instructions of 64 x 16 B; wave wq of the stage's group issues KS of them.  Slot s = (instruction, lane)
    // holds chunk (s % nch) ^ swz(key) of key s / nch.  The lane's source offsets inside a stage are the same for every stage
    // (registers); a stage is a scalar base + these.  Rows past the image's last key up to the next multiple of 16 are read too: the
    // producer of `enc` keeps them finite (enc_pad_rows: zero), their weights are exactly 0.
    const char* ebase = (const char*)(a.enc + (size_t)owner * a.cap * d);
    const int swz_mask = (nch < 16 ? nch : 16) - 1;
    unsigned coff[KS];
#pragma unroll
    for (int c = 0; c < KS; ++c) {
        const int inst = wq * KS + c, s = inst * 64 + lane;
        const int key = s / nch, jj = s - key * nch;
        coff[c] = (unsigned)(key * d * 2 + ((jj ^ (xa_swz(key) & swz_mask)) << 4));
    }
    const mg_lds_t ring_lds = mg_lds_addr(ring);
    auto issue = [&](int it, int slot_it) {          // this group's stage of iteration `it` -> ring slot (slot_it, gi)
        const mg_lds_t dst = ring_lds + (unsigned)((slot_it * NG + gi) * stage_bytes + wq * KS * 1024);
        const char* src = ebase + (size_t)(st0 + it * NG + gi) * (size_t)stage_bytes;
        if (a.nt) { for (int c = 0; c < KS; ++c) glds16_async_sv_nt(src, coff[c], dst + c * 1024); }
        else { for (int c = 0; c < KS; ++c) glds16_async_sv(src, coff[c], dst + c * 1024); }
    };
    // LDS byte offsets inside a stage: score operand (key = lane % 16, chunks fb/8 + 4 ks + g) and transposed operand (key 4 g + a,
    // features fb + 32 (t / 2) + 8 b + 4 (t % 2) .. + 4, a = (lane % 16) / 4, b = lane % 4: result row 4 g' + i of tile t = feature
    // fb + 32 (t / 2) + 8 g' + 4 (t % 2) + i).  The swizzle is an XOR on the low 4 bits of the chunk index: the
    // k-step / tile index changes those bits with period 4 / 8 and adds 256 B beyond - 4 + 8 offsets in registers, immediates for the rest.
    constexpr int NSA = KS < 4 ? KS : 4, NTA = NF < 8 ? NF : 8;
    unsigned sa[NSA], ta[NTA];
    {
        const int sx = xa_swz(head) & swz_mask;
#pragma unroll
        for (int k = 0; k < NSA; ++k) sa[k] = (unsigned)(head * d * 2 + ((((fb >> 3) + g + 4 * k) ^ sx) << 4));
        const int ra = (lane & 15) >> 2, rb = lane & 3, tkey = 4 * g + ra, tx = xa_swz(tkey) & swz_mask;
#pragma unroll
        // (tiles 2p, 2p + 1 take the low / high 8 bytes of chunks fb/8 + 4p + b: a lane of the result then holds 8 consecutive features)
        for (int k = 0; k < NTA; ++k) ta[k] = (unsigned)(tkey * d * 2 + ((((fb >> 3) + 4 * (k >> 1) + rb) ^ tx) << 4) + (k & 1) * 8);
    }

    f32x4 acc[NF];
#pragma unroll
    for (int t = 0; t < NF; ++t) acc[t] = acc4_zero();
    // running maximum in log2 units (scores are scaled by qs * log2(e), weights are exp2), stale by up to XA_DEFER: the accumulators are
    // rescaled only when some head's maximum grew by more than that (weights stay below 2^XA_DEFER; fp32 sums, bf16 weights: no loss)
    float mrun = XA_NEG, lsum = 0.f;
    const float qs2 = qs * 1.44269504088896341f;

    // retire the prologue's loads by hand: from here on the vector-memory queue holds only the counted copies
    MG_WAIT_VMCNT(0);
    uint4 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        MG_TIE(qr[ks]);
        qf[ks] = head < H ? raw16_get(qr[ks]) : make_uint4(0, 0, 0, 0);
    }
    {
        const int lead = R - 1;        // iterations in flight ahead of the one being consumed
        for (int it = 0; it < lead && it < nst_g; ++it) issue(it, it);
        int slot = 0, slot_lead = lead % R;
        for (int it = 0; it < nit; ++it) {
            // own copies of this iteration's stage have landed when at most those of the later ones in flight remain outstanding
            // (a group that has run out of stages has nothing outstanding and only keeps the barriers company)
            {
                int later = nst_g - 1 - it;
                later = later < lead - 1 ? later : lead - 1;
                xa_wait_copies<KS>(later);
            }
            MG_BARRIER_RAW();          // everybody's copies of the iteration have landed; everybody is done with the previous one (its slots are free)
            const bool mine = it < nst_g;
            if (it + lead < nst_g) issue(it + lead, slot_lead);
            const char* stg = ring + (size_t)(slot * NG + gi) * stage_bytes;
            slot = slot + 1 == R ? 0 : slot + 1;
            slot_lead = slot_lead + 1 == R ? 0 : slot_lead + 1;
            // scores, partial over this wave's features (two accumulators: even / odd k-steps)
            f32x4 s0 = acc4_zero(), s1 = acc4_zero();
            if (mine) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const uint4 e = ld16(stg + sa[ks & 3] + (ks >> 2) * 256);
                    if (ks & 1) s1 = mfma16(e, qf[ks], s1); else s0 = mfma16(e, qf[ks], s0);
                }
                float4* rp = (float4*)red + (size_t)wq * 64 + lane;
                *rp = make_float4(s0[0] + s1[0], s0[1] + s1[1], s0[2] + s1[2], s0[3] + s1[3]);
            }
            MG_WAIT_LGKM0();           // (a __syncthreads() would also drain the copies in flight)
            MG_BARRIER_RAW();
            if (!mine) continue;
            float sc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                const float4 v = ((const float4*)red)[(size_t)ww * 64 + lane];
                sc[0] += v.x; sc[1] += v.y; sc[2] += v.z; sc[3] += v.w;
            }
            // online softmax over the stage's 16 keys; this lane: keys k0 + 4 g + {0..3} of head `head`
#pragma unroll
            for (int e = 0; e < 4; ++e) sc[e] *= qs2;
            const int sti = st0 + it * NG + gi;
            if (sti + 1 == nst_all) {        // only an image's last stage holds keys past its end
                const int kbase = sti * XA_KEYS + 4 * g;
#pragma unroll
                for (int e = 0; e < 4; ++e) sc[e] = (kbase + e < nkeys) ? sc[e] : XA_NEG;
            }
            float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
            mx = fmaxf(mx, lane_xor<16>(mx, lane));
            mx = fmaxf(mx, lane_xor<32>(mx, lane));
            if (wave_any(mx > mrun + XA_DEFER)) {          // (same values in every wave of the group: same decision)
                const float mn = fmaxf(mrun, mx);
                const float al = fast_exp2(mrun - mn);
                mrun = mn;
                lsum *= al;
#pragma unroll
                for (int t = 0; t < NF; ++t) {
                    acc[t][0] *= al; acc[t][1] *= al; acc[t][2] *= al; acc[t][3] *= al;
                }
            }
            const uint2 pt = make_uint2(pack_bf16(fast_exp2(sc[0] - mrun), fast_exp2(sc[1] - mrun)),
                                        pack_bf16(fast_exp2(sc[2] - mrun), fast_exp2(sc[3] - mrun)));
            lsum += (bf16lo(pt.x) + bf16hi(pt.x)) + (bf16lo(pt.y) + bf16hi(pt.y));      // the rounded weights, as the product sees them
#pragma unroll
            for (int t = 0; t < NF; ++t) {
                const uint2 e = lds_read_tr16(stg + ta[t & 7] + (t >> 3) * 256);
                acc[t] = mfma16k16(e, pt, acc[t]);
            }
        }
    }
    // merge the wave groups in group order (group 0 += group 1 ...) through the ring's memory: [wq][t][lane] float4 + (m, l) per lane
    if constexpr (NG > 1) {
        float4* mb = (float4*)smem;
        float2* mm = (float2*)(smem + (size_t)NW * NF * 64 * 16);
        for (int gsrc = 1; gsrc < NG; ++gsrc) {
            MG_WAIT_LGKM0();
            MG_BARRIER_RAW();          // the stages (or the previous group's block) have been read by everybody
            if (gi == gsrc) {
#pragma unroll
                for (int t = 0; t < NF; ++t) mb[((size_t)wq * NF + t) * 64 + lane] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
                mm[(size_t)wq * 64 + lane] = make_float2(mrun, lsum);
            }
            MG_WAIT_LGKM0();
            MG_BARRIER_RAW();
            if (gi == 0) {
                const float2 o = mm[(size_t)wq * 64 + lane];
                const float M = fmaxf(mrun, o.x);
                const float f0 = fast_exp2(mrun - M), f1 = fast_exp2(o.x - M);
                mrun = M;
                lsum = lsum * f0 + o.y * f1;
#pragma unroll
                for (int t = 0; t < NF; ++t) {
                    const float4 v = mb[((size_t)wq * NF + t) * 64 + lane];
                    acc[t][0] = acc[t][0] * f0 + v.x * f1; acc[t][1] = acc[t][1] * f0 + v.y * f1;
                    acc[t][2] = acc[t][2] * f0 + v.z * f1; acc[t][3] = acc[t][3] * f0 + v.w * f1;
                }
            }
        }
        if (gi != 0) return;
    }
    // a lane's l covers its own 4 keys per stage: complete it over the lane groups; write the split's context NORMALISED by its own l as
    // bf16 (with one split this is the value the contraction consumes: same bits as normalising there, half the bytes) and (m, l)
    lsum += lane_xor<16>(lsum, lane);
    lsum += lane_xor<32>(lsum, lane);
    if (head < H) {
        const size_t pi = ((size_t)row * a.nsplit + split) * H + head;
        const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
        uint16_t* pp = a.part + pi * d + fb + 8 * g;
#pragma unroll
        for (int p2 = 0; p2 < NF / 2; ++p2)
            st16(pp + 32 * p2, make_uint4(pack_bf16(acc[2 * p2][0] * inv, acc[2 * p2][1] * inv), pack_bf16(acc[2 * p2][2] * inv, acc[2 * p2][3] * inv),
                                          pack_bf16(acc[2 * p2 + 1][0] * inv, acc[2 * p2 + 1][1] * inv), pack_bf16(acc[2 * p2 + 1][2] * inv, acc[2 * p2 + 1][3] * inv)));
        if (wq == 0 && g == 0) { a.ml[pi * 2] = mrun; a.ml[pi * 2 + 1] = lsum; }
    }
}

// ---- ctx_h = (merged c_h / l) · Wv_h^T ------------------------------------------------------------------------------------
// grid (H, ceil(rows / 16)), 4 waves splitting the d features (k-steps w, w + 4, ...), partial tiles summed through LDS in wave order.
// wv: fragment order [H][d/32 k-steps][4 tiles][64 lanes][8]: lane (m, g) of tile tl holds Wv[h*64 + j(m, tl)][32 ks + 8 g .. + 8],
// j(m, tl) = 32 (tl / 2) + 8 (m / 4) + 4 (tl % 2) + m % 4 - the row interleave that leaves a lane 8 consecutive context columns.
__global__ __launch_bounds__(256) void xctx_contract_kernel(XAttnArgs a) {
    MG_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int h = blockIdx.x, r0 = blockIdx.y * 16;
    const int m = lane & 15, g = lane >> 4;
    const int d = a.d, H = a.H, KT = d >> 5, NS = a.nsplit;
    int row = r0 + m;
    const bool in = row < a.rows;
    row = in ? row : a.rows - 1;
    // merge weights of the key splits (each stored normalised by its own l): c = sum_s exp2(m_s - M) l_s c_s / sum_s exp2(m_s - M) l_s
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    if (NS > 1) {
        float ms[4], ls[4], M = XA_NEG;
        for (int s = 0; s < NS; ++s) {
            const size_t pi = ((size_t)row * NS + s) * H + h;
            ms[s] = a.ml[pi * 2]; ls[s] = a.ml[pi * 2 + 1];
            M = fmaxf(M, ms[s]);
        }
        float L = 0.f;
        for (int s = 0; s < NS; ++s) { cs[s] = fast_exp2(ms[s] - M) * ls[s]; L += cs[s]; }      // (the stream's maxima are in log2 units)
        const float inv = L > 0.f ? 1.0f / L : 0.f;
        for (int s = 0; s < NS; ++s) cs[s] *= inv;
    }
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = acc4_zero();
    for (int ks = w; ks < KT; ks += 4) {
        uint4 cb;
        if (NS == 1) {
            cb = ld16_stream(a.part + ((size_t)row * H + h) * d + 32 * ks + 8 * g);          // one split: already the normalised bf16 context (read once)
        } else {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < NS; ++s) {
                const uint4 x = ld16(a.part + (((size_t)row * NS + s) * H + h) * d + 32 * ks + 8 * g);
                v[0] += cs[s] * bf16lo(x.x); v[1] += cs[s] * bf16hi(x.x); v[2] += cs[s] * bf16lo(x.y); v[3] += cs[s] * bf16hi(x.y);
                v[4] += cs[s] * bf16lo(x.z); v[5] += cs[s] * bf16hi(x.z); v[6] += cs[s] * bf16lo(x.w); v[7] += cs[s] * bf16hi(x.w);
            }
            cb = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
        }
        const uint16_t* wp = a.wv + (((size_t)h * KT + ks) * 4) * 512 + lane * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(ld16(wp + t * 512), cb, acc[t]);
    }
    float4* red = (float4*)smem;          // [4 waves][4 tiles][64]
#pragma unroll
    for (int t = 0; t < 4; ++t) red[((size_t)w * 4 + t) * 64 + lane] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    __syncthreads();
    if (w == 0) {
        float o[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            o[t][0] = o[t][1] = o[t][2] = o[t][3] = 0.f;
            for (int ww = 0; ww < 4; ++ww) {
                const float4 x = red[((size_t)ww * 4 + t) * 64 + lane];
                o[t][0] += x.x; o[t][1] += x.y; o[t][2] += x.z; o[t][3] += x.w;
            }
        }
        if (in) {
            const int ld = a.ctx_ld ? a.ctx_ld : H * 64;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)        // tiles 2 pr, 2 pr + 1: columns h*64 + 32 pr + 8 g + {0..3}, {4..7}
                st16(a.ctx + pk_off(row, a.ctx_col0 + h * 64 + 32 * pr + 8 * g, ld),
                     make_uint4(pack_bf16(o[2 * pr][0], o[2 * pr][1]), pack_bf16(o[2 * pr][2], o[2 * pr][3]),
                                pack_bf16(o[2 * pr + 1][0], o[2 * pr + 1][1]), pack_bf16(o[2 * pr + 1][2], o[2 * pr + 1][3])));
        }
    }
}

// ---- encoder states as the stream reads them --------------------------------------------------------------------------------
// src: packed fragment tiles [rows_src][d] (the final norm's bf16 output, or the packed e1 tokens); row_map[r] = row of the image's
// stream (compacted to attended positions, e1 tokens first) or -1; rows_per_image source rows per image; dst [B][cap][d] natural rows.
__global__ __launch_bounds__(256) void enc_rows_kernel(const uint16_t* src, const int* row_map, uint16_t* dst, int B, int rows_per_image, int cap, int d) {
    const int nch = d >> 3;
    const size_t total = (size_t)B * rows_per_image * nch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nch);
        const size_t r = i / nch;
        const int b = (int)(r / rows_per_image);
        const int to = row_map[r];
        if (to < 0) continue;
        st16(dst + ((size_t)b * cap + to) * d + c * 8, ld16(src + pk_off((int)r, c * 8, d)));
    }
}

// rows [len[b], next multiple of 16) of every image's stream cleared: the stream kernel reads whole stages of 16 keys (zero weight past the end)
__global__ __launch_bounds__(256) void enc_pad_rows_kernel(uint16_t* dst, const int* len, int cap, int d) {
    const int b = blockIdx.x, n = len[b];
    int end = (n + XA_KEYS - 1) / XA_KEYS * XA_KEYS;
    end = end < cap ? end : cap;
    const int nch = d >> 3, total = (end - n) * nch;
    for (int i = threadIdx.x; i < total; i += blockDim.x)
        st16(dst + ((size_t)b * cap + n) * d + (size_t)i * 8, make_uint4(0, 0, 0, 0));
}
void enc_pad_rows(uint16_t* dst, const int* len, int B, int cap, int d, mgStream_t stream) {
    MG_LAUNCH(enc_pad_rows_kernel, dim3(B), dim3(256), 0, stream, dst, len, cap, d);
}

// absorbed weights of one decoder layer from the fp32 row-major cross K/V weight [2 * inner][d] (K rows first)
__global__ __launch_bounds__(256) void xattn_pack_weights_kernel(const float* wkv, uint16_t* wk, uint16_t* wv, int H, int d) {
    const int inner = H * 64, KT = d >> 5;
    const size_t n = (size_t)H * d * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        {   // wk[h][f][j] = Wk[h*64 + j][f]
            const int j = (int)(i & 63);
            const size_t hf = i >> 6;
            const int f = (int)(hf % d), h = (int)(hf / d);
            wk[i] = f32_to_bf16(wkv[((size_t)h * 64 + j) * d + f]);
        }
        {   // wv[h][ks][tile][lane][e] = Wv[h*64 + j(lane % 16, tile)][32 ks + 8 (lane / 16) + e]
            const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), tl = (int)((i >> 9) & 3);
            const size_t hk = i >> 11;
            const int ks = (int)(hk % KT), h = (int)(hk / KT);
            const int mm = lane & 15, gg = lane >> 4;
            const int j = 32 * (tl >> 1) + 8 * (mm >> 2) + 4 * (tl & 1) + (mm & 3);
            wv[i] = f32_to_bf16(wkv[((size_t)inner + h * 64 + j) * d + 32 * ks + 8 * gg + e]);
        }
    }
}

void xattn_pack_weights(const float* wkv_f32, uint16_t* wk, uint16_t* wv, int H, int d, mgStream_t stream) {
    MG_LAUNCH(xattn_pack_weights_kernel, dim3(512), dim3(256), 0, stream, wkv_f32, wk, wv, H, d);
}

void enc_rows(const uint16_t* src_pk, const int* row_map, uint16_t* dst, int B, int rows_per_image, int cap, int d, mgStream_t stream) {
    const size_t total = (size_t)B * rows_per_image * (d >> 3);
    size_t blocks = (total + 255) / 256;
    blocks = blocks < 4096 ? blocks : 4096;
    MG_LAUNCH(enc_rows_kernel, dim3((unsigned)(blocks ? blocks : 1)), dim3(256), 0, stream, src_pk, row_map, dst, B, rows_per_image, cap, d);
}

// f-tiles (16 features) per wave of the stream kernel for a model width; NW = d / (16 NF) waves.  0 = width not supported.
int xattn_nf(int d) {
    if (d == 64) return 2;                                                   // 2 waves x 32 features
    const int nf = d / 64;                                                   // 4 waves x d / 4 features
    if (d % 64 == 0 && (nf == 2 || nf == 4 || nf == 8 || nf == 12 || nf == 16)) return nf;
    return 0;
}
bool xattn_supported(int d, int H) { return H >= 1 && H <= 16 && xattn_nf(d) > 0; }

// Two forms of the stream kernel, chosen by the ring: nstg = 4 -> two wave groups (a ring of two iterations x two stages, 136 KB of LDS:
// fastest per workgroup, but nothing else that needs LDS fits beside it on the CU); nstg = 3 -> one wave group, three stages (100 KB: a
// workgroup of the decode projections - 33 KB - can share the CU).
static int xa_groups(int nstg) { return (nstg % 2 == 0) ? 2 : 1; }
size_t xattn_stream_lds(int d, int nstg) {
    const int nf = xattn_nf(d), nw = nf ? d / (16 * nf) : 1;
    return (size_t)nstg * XA_KEYS * d * 2 + (size_t)xa_groups(nstg) * nw * 64 * 16;       // ring + score partials
}

void xattn_expand(const XAttnArgs& a, mgStream_t stream) {
    const int nw = a.d >= 128 ? 4 : a.d / 32;
    MG_LAUNCH(xq_expand_kernel, dim3(a.H, a.d / (32 * nw)), dim3(nw * 64), 0, stream, a);
}

void xattn_stream(const XAttnArgs& a, mgStream_t stream) {
    const int nf = xattn_nf(a.d), nw = a.d / (16 * nf), ng = xa_groups(a.nstg);
    const dim3 grid(a.rows * a.nsplit), block(nw * ng * 64);
    const size_t sh = xattn_stream_lds(a.d, a.nstg);
#define MG_XS(N, W) if (nf == N && nw == W) { if (ng == 2) MG_LAUNCH((xattn_stream_kernel<N, W, 2>), grid, block, sh, stream, a); \
                                              else MG_LAUNCH((xattn_stream_kernel<N, W, 1>), grid, block, sh, stream, a); return; }
    MG_XS(2, 2) MG_XS(2, 4) MG_XS(4, 4) MG_XS(8, 4) MG_XS(12, 4) MG_XS(16, 4)
#undef MG_XS
}
// the stream kernel's LDS request exceeds the default limit: set once per form, outside any stream capture
void xattn_stream_prepare(int d, int nstg) {
    const int nf = xattn_nf(d), nw = nf ? d / (16 * nf) : 0, ng = xa_groups(nstg);
    const size_t sh = xattn_stream_lds(d, nstg);
    (void)sh; (void)nw; (void)ng;
#define MG_XP(N, W) if (nf == N && nw == W) { if (ng == 2) MG_SET_MAX_SMEM((&xattn_stream_kernel<N, W, 2>), sh); else MG_SET_MAX_SMEM((&xattn_stream_kernel<N, W, 1>), sh); return; }
    MG_XP(2, 2) MG_XP(2, 4) MG_XP(4, 4) MG_XP(8, 4) MG_XP(12, 4) MG_XP(16, 4)
#undef MG_XP
}

void xattn_contract(const XAttnArgs& a, mgStream_t stream) {
    MG_LAUNCH(xctx_contract_kernel, dim3(a.H, (a.rows + 15) / 16), dim3(256), (size_t)4 * 4 * 64 * 16, stream, a);
}

}  // namespace mg
