// Decode-step kernels (one new token per live sequence): single-query attention over the KV streams and greedy
// token selection.  These are the HBM-bound kernels of the path: per step the cross-attention K/V of every image
// (2*N_dec*d*S_x bf16 bytes per image) is streamed exactly once.
#include "mg_kernels.h"
#include <mutex>

namespace mg {

constexpr float DC_NEG = -1.0e30f;

// Single-query attention for G queries that share one K/V stream (G = 1: decoder self-attention over the
// sequence's own cache, stock:470-485 bias; G = num_beams: cross-attention of all beams of one image, so the
// image's K/V is read once per step, not once per beam).
// Layout: K/V rows of 64 bf16 (128 B).  A wave reads 8 keys per load instruction (8 lanes x 16 B per key), U
// instructions in flight for K and V each; scores are reduced over the 8 lanes of a key by xor-shuffles;
// every lane keeps an online-softmax partial (m, l, acc[8 dims]) per query which is merged across key slots by
// shuffles and across the 4 waves through LDS in a fixed order (deterministic).
// XA = cross-attention form (per-image key counts from `len`, no positional bias, no beam ancestor table): a distinct
// symbol, so that kernel traces keep the bandwidth-sized cross-attention apart from the short self-attention launches
template <int G, int NW, bool XA, bool TRACE = false, bool ROPE = false>
__global__ __launch_bounds__(NW * 64) void attn_step_kernel(AttnStepArgs a, long long* trace = nullptr) {
    MG_DYN_SMEM(smem);
    // TRACE (tools/trace_attn.py only): shader-clock stamps of wave phases -> trace[(workgroup*NW + wave)*8 + k]
    auto stamp = [&](int k) {
#ifndef MG_EMU
        if constexpr (TRACE) { if ((threadIdx.x & 63) == 0) trace[((size_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 8 + k] = (long long)__builtin_readcyclecounter(); }
#endif
        (void)k;
    };
    stamp(0);
    // keys per wave per round = 8*U: the 8-wave (long-stream) form uses U = 2 so the last, partially filled round of a
    // ~1000-key stream wastes < 10 % of the wave-rounds instead of ~20 %
    constexpr int U = (NW >= 8) ? 2 : 4;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int sub = lane & 7, ks = lane >> 3;
    const int owner = blockIdx.x / a.H, h = blockIdx.x - owner * a.H;
    if constexpr (G == 1 || ROPE) { if (a.live && a.live[owner] == 0) return; }       // finished / idle row (whole workgroup: uniform)
    else { if (a.live && a.kv_owner && a.live[owner * G] == 0) return; }              // beam queue form: idle image slot (its G rows share the flag)
    int tcur = a.pos_rows ? a.pos_rows[owner] + a.t_off : (a.t_dev ? *a.t_dev + a.t_off : a.t);
    // the K/V stream this row reads (and, for self-attention, appends to): a pool entry of the continuous decoders (cross form: the
    // image's encoder K/V; rotary form: the page's own cache), else the row itself
    const int kvo = a.kv_owner ? a.kv_owner[owner] : owner;
    if constexpr (ROPE) { if (a.t_off_rows) tcur += a.t_off_rows[kvo]; }      // prompts of different lengths: the page's own length - t_off
    const int nkeys_all = XA ? a.len[kvo] : ((a.t_dev || a.pos_rows) ? tcur + 1 : a.n_keys);
    // with an in-kernel append the newest key (position t) comes from registers, the cache holds [0, t)
    const bool app0 = ROPE || ((G == 1) && a.qkv.P && a.self_append);
    const int nkeys = app0 ? nkeys_all - 1 : nkeys_all;

    const int inner = a.H * 64;
    // Sum of the split-K partial slabs for this lane's 8 consecutive columns, rounded to bf16 (the projections'
    // storage precision).  The 8 key-slot groups of the wave each fetch a different slab (one L2 round trip), the
    // partial sums are then combined across the groups by xor-shuffles — every lane ends with the full sum.
    auto slab_chunk = [&](int row, int col) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int s = ks + 8 * rep;
            if (s < a.qkv.KS) {
                const float* p = a.qkv.P + (size_t)s * a.qkv.stride + (size_t)row * a.qkv.ldp + col;
                const float4 pa = *(const float4*)p, pb = *(const float4*)(p + 4);
                v[0] += pa.x; v[1] += pa.y; v[2] += pa.z; v[3] += pa.w; v[4] += pb.x; v[5] += pb.y; v[6] += pb.z; v[7] += pb.w;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = sum_slots(v[j], lane);
        return make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
    };
    uint4 q[G];
    uint4 knew = make_uint4(0, 0, 0, 0), vnew = make_uint4(0, 0, 0, 0);
    if constexpr (ROPE) {
        // rotary form: the workgroup owns (sequence `owner`, key/value head h) and its G = H_q / H_kv query heads h*G .. h*G + G-1
        // (grouped-query attention: repeat_kv is never materialised, the cache holds H_kv heads and is streamed once per group)
        static_assert(!XA, "rotary form is self-attention");
        const float* row = a.rope.qkv + (size_t)owner * a.rope.ld;
        const int Hq = a.H * G;
        float r = 1.0f;
        if (a.rope.rs.part) {        // deferred RMSNorm scale of the row (every wave sums the partials itself, fixed order)
            float t = 0.f;
            for (int i = lane; i < a.rope.rs.nparts; i += 64) t += a.rope.rs.part[(size_t)owner * a.rope.rs.nparts + i];
            t = sum_slots(sum8(t), lane);
            r = rsqrtf(t * a.rope.rs.inv_d + a.rope.rs.eps);
        }
        // this lane's 8 dims d = sub*8 + j pair with d +- 32 (rotate_half); both use the angles (sub & 3)*8 + j
        const float* cs = a.rope.cs + (size_t)tcur * 64 + (sub & 3) * 8;
        const float4 c0 = *(const float4*)cs, c1 = *(const float4*)(cs + 4), s0 = *(const float4*)(cs + 32), s1 = *(const float4*)(cs + 36);
        const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const float sgn = sub < 4 ? -1.0f : 1.0f;
        auto rot8 = [&](const float* base, float scale) {
            const float4 a0 = *(const float4*)(base + sub * 8), a1 = *(const float4*)(base + sub * 8 + 4);
            const float4 b0 = *(const float4*)(base + (sub ^ 4) * 8), b1 = *(const float4*)(base + (sub ^ 4) * 8 + 4);
            const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, y[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = ((x[j] * r) * cv[j] + sgn * (y[j] * r) * sv[j]) * scale;
            return make_uint4(pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7]));
        };
#pragma unroll
        for (int g = 0; g < G; ++g) q[g] = rot8(row + (h * G + g) * 64, a.rope.qscale);
        knew = rot8(row + (Hq + h) * 64, 1.0f);
        const float* vb = row + (Hq + a.H + h) * 64 + sub * 8;
        const float4 v0 = *(const float4*)vb, v1 = *(const float4*)(vb + 4);
        vnew = make_uint4(pack_bf16(v0.x * r, v0.y * r), pack_bf16(v0.z * r, v0.w * r), pack_bf16(v1.x * r, v1.y * r), pack_bf16(v1.z * r, v1.w * r));
        if (w == 0 && ks == 0) {
            const size_t off = (((size_t)kvo * a.H + h) * (size_t)a.cap + (size_t)tcur) * 64 + sub * 8;
            st16(a.Kc_w + off, knew);
            st16(a.Vc_w + off, vnew);
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            int row = owner * G + g;
            row = row < a.rows ? row : a.rows - 1;
            q[g] = a.qkv.P ? slab_chunk(row, h * 64 + sub * 8) : ld16(a.q + ((size_t)row * a.H + h) * 64 + sub * 8);
        }
    }
    // self-attention with split-K projections: this workgroup also owns the new position's k, v (kept in registers
    // for its own use and appended to the cache for later steps)
    const bool append = ROPE || ((G == 1) && a.qkv.P && a.self_append);
    if (!ROPE && append) {
        knew = slab_chunk(owner, inner + h * 64 + sub * 8);
        vnew = slab_chunk(owner, 2 * inner + h * 64 + sub * 8);
        if (w == 0 && ks == 0) {
            const size_t off = (((size_t)owner * a.H + h) * (size_t)a.cap + (size_t)tcur) * 64 + sub * 8;
            st16(a.Kc_w + off, knew);
            st16(a.Vc_w + off, vnew);
        }
    }
    float m[G], l[G], acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = DC_NEG;
        l[g] = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[g][d] = 0.f;
    }

    // software-pipelined stream: the loads of the next 8*U keys are issued before the current ones are consumed
    // (self-attention: the positional bias of a key travels with its K/V loads - a load issued where it is consumed
    //  exposes a whole L2 round trip per round)
    auto issue = [&](int kb, uint4 (&kv)[U], uint4 (&vv)[U], float (&bb)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int kc = kb + u * 8 + ks;
            kc = kc < nkeys ? kc : nkeys - 1;
            const int prow = (!XA && a.anc) ? a.anc[(size_t)kc * a.rows + owner] : kvo;
            const size_t off = (((size_t)prow * a.H + h) * (size_t)a.cap + (size_t)kc) * 64 + sub * 8;
            kv[u] = (NW >= 8) ? ld16_stream(a.Kc + off) : ld16(a.Kc + off);
            vv[u] = (NW >= 8) ? ld16_stream(a.Vc + off) : ld16(a.Vc + off);
            bb[u] = 0.f;
            if (!XA && a.bias) {
                int dist = tcur - kc;
                dist = dist < 0 ? 0 : dist;
                bb[u] = a.bias[(size_t)dist * a.H + h];
            }
        }
    };
    uint4 kn[U], vn[U];
    float bn[U];
    if (w * 8 * U < nkeys) issue(w * 8 * U, kn, vn, bn);
    stamp(1);
    // (after the first K/V round is in flight: the partial-sum round trip below overlaps it)
    // deferred RMSNorm of the query rows: q was projected from the un-normalised bf16(h), the row scale r(row) is applied
    // to the scores (q·k is linear in q); every wave sums the row's partials itself (fixed order: deterministic)
    float qs[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        qs[g] = 1.0f;
        if (a.qrs.part) {
            int row = owner * G + g;
            row = row < a.rows ? row : a.rows - 1;
            float t = 0.f;
            for (int i = lane; i < a.qrs.nparts; i += 64) t += a.qrs.part[(size_t)row * a.qrs.nparts + i];
            t = sum_slots(sum8(t), lane);
            qs[g] = rsqrtf(t * a.qrs.inv_d + a.qrs.eps);
        }
    }
    for (int kb = w * 8 * U; kb < nkeys; kb += NW * 8 * U) {
        uint4 kv[U], vv[U];
        float bcur[U];
        int key[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { kv[u] = kn[u]; vv[u] = vn[u]; bcur[u] = bn[u]; key[u] = kb + u * 8 + ks; }
        if (kb + NW * 8 * U < nkeys) issue(kb + NW * 8 * U, kn, vn, bn);
        float s[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float bias = bcur[u];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float p = dot2_bf16(q[g].x, kv[u].x, 0.f);
                p = dot2_bf16(q[g].y, kv[u].y, p);
                p = dot2_bf16(q[g].z, kv[u].z, p);
                p = dot2_bf16(q[g].w, kv[u].w, p);
                p = sum8(p);
                s[u][g] = key[u] < nkeys ? p * qs[g] + bias : DC_NEG;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float mx = s[0][g];
#pragma unroll
            for (int u = 1; u < U; ++u) mx = fmaxf(mx, s[u][g]);
            const float mn = fmaxf(m[g], mx);
            const float al = fast_exp(m[g] - mn);
            m[g] = mn;
            l[g] *= al;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[g][d] *= al;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float p = key[u] < nkeys ? fast_exp(s[u][g] - mn) : 0.f;
                l[g] += p;
                acc[g][0] += p * bf16lo(vv[u].x); acc[g][1] += p * bf16hi(vv[u].x);
                acc[g][2] += p * bf16lo(vv[u].y); acc[g][3] += p * bf16hi(vv[u].y);
                acc[g][4] += p * bf16lo(vv[u].z); acc[g][5] += p * bf16hi(vv[u].z);
                acc[g][6] += p * bf16lo(vv[u].w); acc[g][7] += p * bf16hi(vv[u].w);
            }
        }
        if constexpr (TRACE) { if (kb == w * 8 * U) stamp(2); }
    }
    if (append && w == 0) {   // the new position (distance 0), handled by key slot 0 of wave 0; whole wave runs the shuffles
#pragma unroll
        for (int g = 0; g < (ROPE ? G : 1); ++g) {
            float p = dot2_bf16(q[g].x, knew.x, 0.f);
            p = dot2_bf16(q[g].y, knew.y, p);
            p = dot2_bf16(q[g].z, knew.z, p);
            p = dot2_bf16(q[g].w, knew.w, p);
            p = sum8(p);
            if (ks == 0) {
                const float sc = p * qs[g] + (a.bias ? a.bias[h] : 0.f);
                const float mn = fmaxf(m[g], sc);
                const float al = fast_exp(m[g] - mn), pe = fast_exp(sc - mn);
                m[g] = mn;
                l[g] = l[g] * al + pe;
                acc[g][0] = acc[g][0] * al + pe * bf16lo(vnew.x); acc[g][1] = acc[g][1] * al + pe * bf16hi(vnew.x);
                acc[g][2] = acc[g][2] * al + pe * bf16lo(vnew.y); acc[g][3] = acc[g][3] * al + pe * bf16hi(vnew.y);
                acc[g][4] = acc[g][4] * al + pe * bf16lo(vnew.z); acc[g][5] = acc[g][5] * al + pe * bf16hi(vnew.z);
                acc[g][6] = acc[g][6] * al + pe * bf16lo(vnew.w); acc[g][7] = acc[g][7] * al + pe * bf16hi(vnew.w);
            }
        }
    }
    stamp(3);
    // merge the 8 key slots of the wave: a halving reduction over lane bits 5, 4, 3 (m and l travel whole, the accumulators
    // half at a time: 4 + 2 + 1 exchanges instead of 3 x 8), after which lane (ks, sub) holds feature sub*8 + ks of the head
    float o1[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float mo = lane_xor<32>(m[g], lane), lo = lane_xor<32>(l[g], lane);
        float M = fmaxf(m[g], mo);
        float f1 = fast_exp(m[g] - M), f2 = fast_exp(mo - M);
        l[g] = l[g] * f1 + lo * f2;
        m[g] = M;
        float a4[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) a4[d] = lane_halve<32>(acc[g][d], acc[g][d + 4], f1, f2, lane);
        mo = lane_xor<16>(m[g], lane); lo = lane_xor<16>(l[g], lane);
        M = fmaxf(m[g], mo);
        f1 = fast_exp(m[g] - M); f2 = fast_exp(mo - M);
        l[g] = l[g] * f1 + lo * f2;
        m[g] = M;
        const float a20 = lane_halve<16>(a4[0], a4[2], f1, f2, lane), a21 = lane_halve<16>(a4[1], a4[3], f1, f2, lane);
        mo = lane_xor<8>(m[g], lane); lo = lane_xor<8>(l[g], lane);
        M = fmaxf(m[g], mo);
        f1 = fast_exp(m[g] - M); f2 = fast_exp(mo - M);
        l[g] = l[g] * f1 + lo * f2;
        m[g] = M;
        o1[g] = lane_halve<8>(a20, a21, f1, f2, lane);
    }
    stamp(4);
    // merge the NW waves through LDS: red[(w*G + g)*66 + feature], m at +64, l at +65 (fixed order: deterministic)
    float* red = (float*)smem;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float* r = red + ((size_t)w * G + g) * 66;
        r[sub * 8 + ks] = o1[g];
        if (lane == 0) { r[64] = m[g]; r[65] = l[g]; }
    }
    __syncthreads();
    stamp(5);
    if (w == 0) {          // lane = feature of the head
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float mw[NW];
            float M = DC_NEG;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) { mw[ww] = red[((size_t)ww * G + g) * 66 + 64]; M = fmaxf(M, mw[ww]); }
            float L = 0.f, o = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                const float* r = red + ((size_t)ww * G + g) * 66;
                const float f = fast_exp(mw[ww] - M);
                L += r[65] * f;
                o += r[lane] * f;
            }
            const float inv = L > 0.f ? 1.0f / L : 0.f;
            if constexpr (ROPE) {      // query head h*G + g of sequence `owner`
                a.ctx[pk_off(owner, a.ctx_col0 + (h * G + g) * 64 + lane, a.ctx_ld ? a.ctx_ld : a.H * G * 64)] = f32_to_bf16_rn(o * inv);
            } else {
                const int row = owner * G + g;
                if (row < a.rows)
                    a.ctx[pk_off(row, a.ctx_col0 + h * 64 + lane, a.ctx_ld ? a.ctx_ld : a.H * 64)] = f32_to_bf16_rn(o * inv);
            }
        }
    }
    stamp(6);
}

#ifdef MG_TOOLS
// instrumented cross-attention (G = 1), tools/trace_attn.py
void attention_step_trace(const AttnStepArgs& a, long long* trace, mgStream_t stream) {
    const dim3 grid(a.rows * a.H), block(512);
    MG_LAUNCH((attn_step_kernel<1, 8, true, true>), grid, block, (size_t)8 * 8 * 10 * sizeof(float), stream, a, trace);
}
#endif

constexpr size_t AS_SHARED_LDS = (size_t)84 * 1024;      // more than half a CU's 160 KB: one resident workgroup per CU
// the kernel attribute that permits the large LDS request, set once and outside any stream capture (mg_set_shared_gpu)
void attention_step_allow_shared() {
    static std::once_flag once;
    std::call_once(once, [] { MG_SET_MAX_SMEM((&attn_step_kernel<1, 8, true>), AS_SHARED_LDS); });
}

void attention_step(const AttnStepArgs& a, mgStream_t stream) {
    const int G = a.group;
    if (a.rope.qkv) {            // rotary self-attention step: H = key/value heads, group = query heads per key/value head
        const dim3 rgrid(a.rows * a.H);
        const size_t rsh = (size_t)8 * G * 8 * 10 * sizeof(float);
#define MG_AR(GG) case GG: MG_LAUNCH((attn_step_kernel<GG, 8, false, false, true>), rgrid, dim3(8 * 64), rsh, stream, a, (long long*)nullptr); break;
        switch (G) { MG_AR(1) MG_AR(2) MG_AR(3) MG_AR(4) MG_AR(6) MG_AR(8) default: break; }
#undef MG_AR
        return;
    }
    const int owners = (a.rows + G - 1) / G;
    const dim3 grid(owners * a.H);
    // 8 waves per (row or image, head) - 128 keys per round, more loads in flight per CU - unless the grid alone fills
    // the chip with the short self-attention streams (beam search), where 4 waves win.  Measured end to end: greedy B=32
    // (512 workgroups) 8 waves +1.2 % over 4, 16 waves -3 %; beam-5 (2560 workgroups) 4 waves +4 % over 8.
    // The choice must not depend on the number of rows in the call: the waves partition the keys, so 4 and 8 waves merge
    // their online-softmax partials in different orders and a row's bits would change with its batch size (64 greedy rows
    // used to switch to 4 waves: a row decoded in a 64-row call differed from the same row in a 32-row call).  Beam rows
    // (ancestor table) take 4 waves, everything else 8.
    const bool eight = a.len != nullptr || a.anc == nullptr;
    const int NW = eight ? 8 : 4;
    const dim3 block(NW * 64);
    size_t sh = (size_t)NW * G * 8 * 10 * sizeof(float);
    // Shared GPU (several execution contexts in flight, mg_set_shared_gpu): the K/V stream of the cross-attention is the one launch of
    // a decode step whose workgroups live long (a workgroup streams its (image, head)'s 268 KB) and whose grid fills every wave slot of
    // the chip (4 workgroups of 8 waves per CU).  The latency-sized launches of the OTHER contexts then queue for wave slots behind it
    // (measured at 128 rows, 4 contexts: the QKV projection 13 us alone, 82 us in flight).  One resident workgroup per CU still streams at
    // 5.9 TB/s alone (8 waves x 8 loads in flight per lane) and leaves three quarters of the wave slots to the others: 143.2 -> 148.1
    // images/s with four contexts, 107 -> 103 for a call alone - hence per context and off by default.  The residency is capped through
    // the LDS request (more than half a CU's 160 KB); same kernel, same bits.
    if (a.one_wg_per_cu && a.len && G == 1 && sh < AS_SHARED_LDS) sh = AS_SHARED_LDS;      // (attention_step_allow_shared ran when the setting was made)
#define MG_AS(GG)                                                                                         \
    case GG:                                                                                              \
        if (a.len) MG_LAUNCH((attn_step_kernel<GG, 8, true>), grid, block, sh, stream, a, (long long*)nullptr);               \
        else if (eight) MG_LAUNCH((attn_step_kernel<GG, 8, false>), grid, block, sh, stream, a, (long long*)nullptr);          \
        else MG_LAUNCH((attn_step_kernel<GG, 4, false>), grid, block, sh, stream, a, (long long*)nullptr);                     \
        break;
    switch (G) {
        MG_AS(1) MG_AS(2) MG_AS(3) MG_AS(4) MG_AS(5) MG_AS(6) MG_AS(7) MG_AS(8)
        default: break;
    }
#undef MG_AS
}

// Greedy selection (gen:2925-2937): argmax with lowest-index tie-break (torch.argmax), finished rows emit pad,
// EOS bookkeeping.  One workgroup of 1024 threads per row, 16-byte loads.
constexpr int GS_THREADS = 1024;
MG_DEV void top2_merge(float& b1, float& b2, int& i1, float o1, float o2, int oi) {
    if (o1 > b1 || (o1 == b1 && oi < i1)) { b2 = fmaxf(b1, o2); b1 = o1; i1 = oi; }
    else b2 = fmaxf(b2, o1);
}
__global__ __launch_bounds__(GS_THREADS) void greedy_select_kernel(ArgmaxArgs a) {
    MG_DYN_SMEM(smem);
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* lg = a.logits + (size_t)row * a.ldl;
    const bool stream = a.slots.pos != nullptr;
    const int pos = stream ? a.slots.pos[row] + 1 : (a.pos_dev ? *a.pos_dev + a.pos : a.pos);     // column written
    const bool no_eos = a.suppress_eos || pos < a.min_len;
    // stop tokens in registers (the argument block lives in device memory: no loads from it inside the scan)
    const int e0 = a.eos, e1 = a.n_eos_more > 0 ? a.eos_more[0] : -1, e2 = a.n_eos_more > 1 ? a.eos_more[1] : -1,
              e3 = a.n_eos_more > 2 ? a.eos_more[2] : -1;
    const bool one_eos = a.n_eos_more == 0;
    auto is_eos = [&](int i) { return i == e0 || (!one_eos && (i == e1 || i == e2 || i == e3)); };
    float b1 = -3.0e38f, b2 = -3.0e38f;
    int i1 = 0x7fffffff;
    const int nq = (a.V + 3) >> 2;      // rows are padded to a multiple of 32 floats, so the last float4 is readable
    // batches of 4 independent 16-byte loads per thread: one L2 round trip per batch instead of one per load (a row of
    // 33 201 logits is 8.1 loads per thread; the kernel is latency-, not bandwidth-sized)
    for (int c0 = tid; c0 < nq; c0 += 4 * GS_THREADS) {
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * GS_THREADS;
            q[u] = c < nq ? *(const float4*)(lg + c * 4) : make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * GS_THREADS;
            const float vv[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = c * 4 + j;
                float v = vv[j];
                if (i >= a.V || (no_eos && is_eos(i))) v = -3.0e38f;
                if (v > b1 || (v == b1 && i < i1)) { b2 = b1; b1 = v; i1 = i; }
                else if (v > b2) b2 = v;
            }
        }
    }
#pragma unroll
    for (int step = 1; step < 64; step <<= 1) {
        const float o1 = __shfl_xor(b1, step), o2 = __shfl_xor(b2, step);
        const int oi = __shfl_xor(i1, step);
        top2_merge(b1, b2, i1, o1, o2, oi);
    }
    float* rv = (float*)smem;                 // [16][2]
    int* ri = (int*)(smem + 128);             // [16]
    if (lane == 0) { rv[w * 2] = b1; rv[w * 2 + 1] = b2; ri[w] = i1; }
    __syncthreads();
    if (tid == 0) {
        for (int ww = 1; ww < GS_THREADS / 64; ++ww) top2_merge(b1, b2, i1, rv[ww * 2], rv[ww * 2 + 1], ri[ww]);
        if (stream) {
            // continuous decoding: the row writes column `pos` of ITS image; a row that ends frees the slot (slot_refill hands
            // it the next image).  Idle slots computed on stale inputs: nothing is written for them.
            if (a.unfinished[row]) {
                const int img = a.slots.img[row];
                const int64_t tok = (int64_t)i1;
                a.next_ids[row] = tok;
                a.slots.pos[row] = pos;
                if (pos < a.max_len) a.out_ids[(size_t)img * a.max_len + pos] = tok;
                if (is_eos((int)tok) || pos + 1 >= a.max_len) {
                    a.unfinished[row] = 0;
                    a.slots.img[row] = -1;
                    a.slots.out_len[img] = pos + 1 < a.max_len ? pos + 1 : a.max_len;
                    atomicAdd(a.slots.ctr + 1, 1);
                }
            }
            return;
        }
        const int unf = a.unfinished[row];
        const int64_t tok = unf ? (int64_t)i1 : (int64_t)a.pad;
        a.next_ids[row] = tok;
        if (pos < a.max_len) a.out_ids[(size_t)row * a.max_len + pos] = tok;
        const int still = unf && !is_eos((int)tok);
        a.unfinished[row] = still;
        if (still) atomicAdd(a.n_unfinished, 1);
        if (a.top2) {
            float* tp = a.top2 + (a.pos_dev ? (size_t)pos * a.rows * 2 : 0);
            tp[row * 2] = b1; tp[row * 2 + 1] = b2;
        }
        if (a.step_ctr) {
            int* c = a.step_ctr;
            __threadfence();
            if (atomicAdd(c + 6, 1) == a.rows - 1) {        // every other workgroup has finished (and read the step index)
                const int unf_total = atomicAdd(a.n_unfinished, 0);
                *a.n_unfinished = 0;       // plain stores: nobody else touches the counters until the next launch
                c[0] = unf_total;
                if (unf_total == 0 && c[1] < 0) c[1] = c[2];
                c[2] += 1;
                c[6] = 0;
            }
        }
    }
}

// one thread walks the slots in order: the assignment of images to slots is deterministic (it does not change any image's
// result - rows are independent - but it keeps runs reproducible to the byte, K/V cache contents included)
__global__ __launch_bounds__(64) void slot_refill_kernel(SlotTable s, int64_t* next_ids, int* unfinished, int rows) {
    // the slots' state is fetched by the whole wave first (one round trip instead of one per slot for the walking thread)
    MG_DYN_SMEM(smem);
    int* s_unf = (int*)smem;
    int* s_img = s_unf + 256;
    for (int r = threadIdx.x; r < rows; r += 64) { s_unf[r] = unfinished[r]; s_img[r] = s.img[r]; }
    __syncthreads();
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int* c = s.ctr;
    int head = c[4], done_add = 0;
    const int ready = c[5];
    int n_live = 0, oldest = 0x7fffffff;
    for (int r = 0; r < rows; ++r) {
        int unf = s_unf[r], img = s_img[r];
        while (!unf && head < ready) {
            const int i = head++;
            int64_t tok = (int64_t)s.start_id;
            if (s.first_tok) {
                // the sequence's first token was selected by its prefill (column 0 is written): it may already be a stop token
                tok = s.first_tok[i];
                const int t = (int)tok;
                bool stop = s.max_len <= 1;
                for (int k = 0; k < s.n_stop; ++k) stop = stop || t == s.stop[k];
                if (stop) { s.out_len[i] = 1; ++done_add; continue; }
            }
            s.img[r] = i;
            s.pool[r] = i % s.pool_cap;
            s.pos[r] = 0;
            next_ids[r] = tok;
            unfinished[r] = 1;
            unf = 1; img = i;
        }
        if (unf) { ++n_live; oldest = img < oldest ? img : oldest; }
    }
    c[4] = head;
    if (done_add) c[1] += done_add;
    c[0] = n_live;
    c[7] = n_live ? oldest : head;      // every image below this index has finished
    c[2] += 1;
}
void slot_refill(const SlotTable& s, int64_t* next_ids, int* unfinished, int rows, mgStream_t stream) {
    MG_LAUNCH(slot_refill_kernel, dim3(1), dim3(64), 512 * sizeof(int), stream, s, next_ids, unfinished, rows);
}

// Fused tail of the greedy decode step: reduce the lm_head launch's per-workgroup top-2 partials (ArgmaxArgs::ptop) instead of
// scanning V logits, do the row's bookkeeping exactly as greedy_select_kernel, then produce the next step's first activations
// for the selected token (embedding row -> h, bf16(RMSNorm(h) * gain) -> x_pk, bf16(h) -> x2 window) - embed_norm_rows' work.
// One workgroup of 256 threads per row; batch mode only (the continuous decoder refills slots between selection and embedding).
__global__ __launch_bounds__(256) void greedy_select_fused_kernel(ArgmaxArgs a) {
    MG_DYN_SMEM(smem);
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int pos = a.pos_dev ? *a.pos_dev + a.pos : a.pos;
    const bool no_eos = a.suppress_eos || pos < a.min_len;
    float b1 = -3.0e38f, b2 = -3.0e38f;
    int i1 = 0x7fffffff;
    const float4* pt = a.ptop + (size_t)row * a.ntiles;
    for (int c0 = tid; c0 < a.ntiles; c0 += 4 * 256) {          // batches of independent loads (one L2 round trip per batch)
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * 256;
            q[u] = c < a.ntiles ? pt[c] : make_float4(-3.0e38f, -3.0e38f, __int_as_float(0x7fffffff), 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) top2_merge(b1, b2, i1, q[u].x, q[u].y, __float_as_int(q[u].z));
    }
    if (tid < 4 && !no_eos) {                                    // the stop tokens rank with everybody else unless suppressed
        const int e = tid == 0 ? a.eos : (tid - 1 < a.n_eos_more ? a.eos_more[tid - 1] : -1);
        if (e >= 0) top2_merge(b1, b2, i1, a.stopv[(size_t)row * 4 + tid], -3.0e38f, e);
    }
#pragma unroll
    for (int step = 1; step < 64; step <<= 1) {
        const float o1 = __shfl_xor(b1, step), o2 = __shfl_xor(b2, step);
        const int oi = __shfl_xor(i1, step);
        top2_merge(b1, b2, i1, o1, o2, oi);
    }
    float* rv = (float*)smem;                 // [4][2]
    int* ri = (int*)(smem + 64);              // [4], then [8] = the selected token
    if (lane == 0) { rv[w * 2] = b1; rv[w * 2 + 1] = b2; ri[w] = i1; }
    __syncthreads();
    if (tid == 0) {
        for (int ww = 1; ww < 4; ++ww) top2_merge(b1, b2, i1, rv[ww * 2], rv[ww * 2 + 1], ri[ww]);
        const int e0 = a.eos, e1 = a.n_eos_more > 0 ? a.eos_more[0] : -1, e2 = a.n_eos_more > 1 ? a.eos_more[1] : -1, e3 = a.n_eos_more > 2 ? a.eos_more[2] : -1;
        const int unf = a.unfinished[row];
        const int64_t tok = unf ? (int64_t)i1 : (int64_t)a.pad;
        a.next_ids[row] = tok;
        ri[8] = (int)tok;
        if (pos < a.max_len) a.out_ids[(size_t)row * a.max_len + pos] = tok;
        const int t = (int)tok;
        const int still = unf && !(t == e0 || t == e1 || t == e2 || t == e3);
        a.unfinished[row] = still;
        if (still) atomicAdd(a.n_unfinished, 1);
        if (a.top2) {
            float* tp = a.top2 + (a.pos_dev ? (size_t)pos * a.rows * 2 : 0);
            tp[row * 2] = b1; tp[row * 2 + 1] = b2;
        }
        if (a.step_ctr) {
            int* c = a.step_ctr;
            __threadfence();
            if (atomicAdd(c + 6, 1) == a.rows - 1) {
                const int unf_total = atomicAdd(a.n_unfinished, 0);
                *a.n_unfinished = 0;
                c[0] = unf_total;
                if (unf_total == 0 && c[1] < 0) c[1] = c[2];
                c[2] += 1;
                c[6] = 0;
            }
        }
    }
    __syncthreads();
    // the next step's embedding + first RMSNorm for this row (embed_norm_rows_kernel, one row): thread c < d/8 owns 8 columns
    const int tok = ri[8], d = a.d, nch = d >> 3;
    const uint16_t* src = a.tok_emb + (size_t)tok * d;
    float ss = 0.f;
    for (int c = tid; c < nch; c += 256) {
        const uint4 t = ld16(src + c * 8);
        const float a0 = bf16lo(t.x), a1 = bf16hi(t.x), a2 = bf16lo(t.y), a3 = bf16hi(t.y);
        const float c0 = bf16lo(t.z), c1 = bf16hi(t.z), c2 = bf16lo(t.w), c3 = bf16hi(t.w);
        ss += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3 + c0 * c0 + c1 * c1 + c2 * c2 + c3 * c3;
    }
    // NOTE: embed_norm_rows_kernel sums a row in ONE wave (lane c % 64 takes chunks c, c + 64, ...; wave_sum).  The same order
    // here: the sum of squares is over bf16 values, so regrouping could change the last bit of r and with it a rounded output.
    float* red = (float*)(smem + 128);        // [256]
    red[tid] = ss;
    __syncthreads();
    if (w == 0) {
        float t = 0.f;
        for (int k = lane; k < 256; k += 64) t += red[k];      // = what lane `lane` of the single-wave form accumulates (d <= 2048)
        t = wave_sum(t);
        if (lane == 0) red[0] = t;
    }
    __syncthreads();
    const float r = rsqrtf(red[0] / (float)d + a.eps);
    for (int c = tid; c < nch; c += 256) {
        const uint4 t = ld16(src + c * 8);
        const float v[8] = {bf16lo(t.x), bf16hi(t.x), bf16lo(t.y), bf16hi(t.y), bf16lo(t.z), bf16hi(t.z), bf16lo(t.w), bf16hi(t.w)};
        const float4 g0 = *(const float4*)(a.gain + c * 8), g1 = *(const float4*)(a.gain + c * 8 + 4);
        *(float4*)(a.h + (size_t)row * d + c * 8) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(a.h + (size_t)row * d + c * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        st16(a.x_pk + pk_off(row, c * 8, d),
             make_uint4(pack_bf16(g0.x * (v[0] * r), g0.y * (v[1] * r)), pack_bf16(g0.z * (v[2] * r), g0.w * (v[3] * r)),
                        pack_bf16(g1.x * (v[4] * r), g1.y * (v[5] * r)), pack_bf16(g1.z * (v[6] * r), g1.w * (v[7] * r))));
        if (a.x2_pk) st16(a.x2_pk + pk_off(row, a.x2_col0 + c * 8, a.x2_ld), t);
    }
}
void greedy_select_fused(const ArgmaxArgs& a, mgStream_t stream) {
    MG_LAUNCH(greedy_select_fused_kernel, dim3(a.rows), dim3(256), 128 + 256 * sizeof(float), stream, a);
}

void greedy_select(const ArgmaxArgs& a, mgStream_t stream) {
    MG_LAUNCH(greedy_select_kernel, dim3(a.rows), dim3(GS_THREADS), 256, stream, a);
}

}  // namespace mg
