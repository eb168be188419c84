// Decode-step kernels (one new token per live sequence): single-query attention over the KV streams and greedy
// token selection.  These are the HBM-bound kernels of the path: per step the cross-attention K/V of every image
// (2*N_dec*d*S_x bf16 bytes per image) is streamed exactly once.
#include "mg_kernels.h"

namespace mg {

constexpr float DC_NEG = -1.0e30f;

// Single-query attention for G queries that share one K/V stream (G = 1: decoder self-attention over the
// sequence's own cache, stock:470-485 bias; G = num_beams: cross-attention of all beams of one image, so the
// image's K/V is read once per step, not once per beam).
// Layout: K/V rows of 64 bf16 (128 B).  A wave reads 8 keys per load instruction (8 lanes x 16 B per key), U
// instructions in flight for K and V each; scores are reduced over the 8 lanes of a key by xor-shuffles;
// every lane keeps an online-softmax partial (m, l, acc[8 dims]) per query which is merged across key slots by
// shuffles and across the 4 waves through LDS in a fixed order (deterministic).
template <int G>
__global__ __launch_bounds__(256) void attn_step_kernel(AttnStepArgs a) {
    MG_DYN_SMEM(smem);
    constexpr int U = 4;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int sub = lane & 7, ks = lane >> 3;
    const int owner = blockIdx.x / a.H, h = blockIdx.x - owner * a.H;
    const int tcur = a.t_dev ? *a.t_dev : a.t;
    const int nkeys = a.len ? a.len[owner] : (a.t_dev ? tcur + 1 : a.n_keys);

    uint4 q[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        int row = owner * G + g;
        row = row < a.rows ? row : a.rows - 1;
        q[g] = ld16(a.q + ((size_t)row * a.H + h) * 64 + sub * 8);
    }
    float m[G], l[G], acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = DC_NEG;
        l[g] = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[g][d] = 0.f;
    }

    for (int kb = w * 8 * U; kb < nkeys; kb += 4 * 8 * U) {
        uint4 kv[U], vv[U];
        int key[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            key[u] = kb + u * 8 + ks;
            const int kc = key[u] < nkeys ? key[u] : nkeys - 1;
            const int prow = a.anc ? a.anc[(size_t)kc * a.rows + owner] : owner;
            const size_t off = (((size_t)prow * a.H + h) * (size_t)a.cap + (size_t)kc) * 64 + sub * 8;
            kv[u] = ld16(a.Kc + off);
            vv[u] = ld16(a.Vc + off);
        }
        float s[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float bias = 0.f;
            if (a.bias) {
                int dist = tcur - key[u];
                dist = dist < 0 ? 0 : dist;
                bias = a.bias[(size_t)dist * a.H + h];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float p = dot2_bf16(q[g].x, kv[u].x, 0.f);
                p = dot2_bf16(q[g].y, kv[u].y, p);
                p = dot2_bf16(q[g].z, kv[u].z, p);
                p = dot2_bf16(q[g].w, kv[u].w, p);
                p += __shfl_xor(p, 1);
                p += __shfl_xor(p, 2);
                p += __shfl_xor(p, 4);
                s[u][g] = key[u] < nkeys ? p + bias : DC_NEG;
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
    }
    // merge the 8 key slots of the wave
#pragma unroll
    for (int step = 8; step <= 32; step <<= 1) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float mo = __shfl_xor(m[g], step), lo = __shfl_xor(l[g], step);
            const float M = fmaxf(m[g], mo);
            const float f1 = fast_exp(m[g] - M), f2 = fast_exp(mo - M);
            l[g] = l[g] * f1 + lo * f2;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[g][d] = acc[g][d] * f1 + __shfl_xor(acc[g][d], step) * f2;
            m[g] = M;
        }
    }
    // merge the 4 waves: red[w][g][sub][10]
    float* red = (float*)smem;
    if (ks == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float* r = red + (((size_t)w * G + g) * 8 + sub) * 10;
            r[0] = m[g]; r[1] = l[g];
#pragma unroll
            for (int d = 0; d < 8; ++d) r[2 + d] = acc[g][d];
        }
    }
    __syncthreads();
    if (w == 0 && ks == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float M = DC_NEG;
            for (int ww = 0; ww < 4; ++ww) M = fmaxf(M, red[(((size_t)ww * G + g) * 8 + sub) * 10]);
            float L = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int ww = 0; ww < 4; ++ww) {
                const float* r = red + (((size_t)ww * G + g) * 8 + sub) * 10;
                const float f = fast_exp(r[0] - M);
                L += r[1] * f;
#pragma unroll
                for (int d = 0; d < 8; ++d) o[d] += r[2 + d] * f;
            }
            const float inv = L > 0.f ? 1.0f / L : 0.f;
            const int row = owner * G + g;
            if (row < a.rows)
                st16(a.ctx + pk_off(row, h * 64 + sub * 8, a.H * 64),
                     make_uint4(pack_bf16(o[0] * inv, o[1] * inv), pack_bf16(o[2] * inv, o[3] * inv),
                                pack_bf16(o[4] * inv, o[5] * inv), pack_bf16(o[6] * inv, o[7] * inv)));
        }
    }
}

void attention_step(const AttnStepArgs& a, mgStream_t stream) {
    const int G = a.group;
    const int owners = (a.rows + G - 1) / G;
    const dim3 grid(owners * a.H), block(256);
    const size_t sh = (size_t)4 * G * 8 * 10 * sizeof(float);
    switch (G) {
        case 1: MG_LAUNCH((attn_step_kernel<1>), grid, block, sh, stream, a); break;
        case 2: MG_LAUNCH((attn_step_kernel<2>), grid, block, sh, stream, a); break;
        case 3: MG_LAUNCH((attn_step_kernel<3>), grid, block, sh, stream, a); break;
        case 4: MG_LAUNCH((attn_step_kernel<4>), grid, block, sh, stream, a); break;
        case 5: MG_LAUNCH((attn_step_kernel<5>), grid, block, sh, stream, a); break;
        case 6: MG_LAUNCH((attn_step_kernel<6>), grid, block, sh, stream, a); break;
        case 7: MG_LAUNCH((attn_step_kernel<7>), grid, block, sh, stream, a); break;
        case 8: MG_LAUNCH((attn_step_kernel<8>), grid, block, sh, stream, a); break;
        default: break;
    }
}

// Greedy selection (gen:2925-2937): argmax with lowest-index tie-break (torch.argmax), finished rows emit pad,
// EOS bookkeeping.  One workgroup per row.
__global__ __launch_bounds__(256) void greedy_select_kernel(ArgmaxArgs a) {
    MG_DYN_SMEM(smem);
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* lg = a.logits + (size_t)row * a.ldl;
    const int pos = a.pos_dev ? *a.pos_dev : a.pos;
    const bool no_eos = a.suppress_eos || pos < a.min_len;
    float b1 = -3.0e38f, b2 = -3.0e38f;
    int i1 = 0x7fffffff;
    for (int i = tid; i < a.V; i += 256) {
        float v = lg[i];
        if (no_eos && i == a.eos) v = -3.0e38f;
        if (v > b1 || (v == b1 && i < i1)) { b2 = b1; b1 = v; i1 = i; }
        else if (v > b2) b2 = v;
    }
#pragma unroll
    for (int step = 1; step < 64; step <<= 1) {
        const float o1 = __shfl_xor(b1, step), o2 = __shfl_xor(b2, step);
        const int oi = __shfl_xor(i1, step);
        if (o1 > b1 || (o1 == b1 && oi < i1)) { b2 = fmaxf(b1, o2); b1 = o1; i1 = oi; }
        else b2 = fmaxf(b2, o1);
    }
    float* rv = (float*)smem;
    int* ri = (int*)(smem + 32);
    if (lane == 0) { rv[w * 2] = b1; rv[w * 2 + 1] = b2; ri[w] = i1; }
    __syncthreads();
    if (tid == 0) {
        for (int ww = 1; ww < 4; ++ww) {
            const float o1 = rv[ww * 2], o2 = rv[ww * 2 + 1];
            const int oi = ri[ww];
            if (o1 > b1 || (o1 == b1 && oi < i1)) { b2 = fmaxf(b1, o2); b1 = o1; i1 = oi; }
            else b2 = fmaxf(b2, o1);
        }
        const int unf = a.unfinished[row];
        const int64_t tok = unf ? (int64_t)i1 : (int64_t)a.pad;
        a.next_ids[row] = tok;
        if (pos < a.max_len) a.out_ids[(size_t)row * a.max_len + pos] = tok;
        const int still = unf && tok != a.eos;
        a.unfinished[row] = still;
        if (still) atomicAdd(a.n_unfinished, 1);
        if (a.top2) { a.top2[row * 2] = b1; a.top2[row * 2 + 1] = b2; }
    }
}

void greedy_select(const ArgmaxArgs& a, mgStream_t stream) {
    mg_memset_async(a.n_unfinished, 0, sizeof(int), stream);
    MG_LAUNCH(greedy_select_kernel, dim3(a.rows), dim3(256), 64, stream, a);
}

}  // namespace mg
