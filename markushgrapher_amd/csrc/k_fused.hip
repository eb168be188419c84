// Decode step, head-owned fusion of the self-attention sub-layer's front half (stock:382-575 with past key/values):
//   q, k, v = Wq/Wk/Wv[head] · RMSNorm(h)      (deferred norm: X = bf16(h * gain), rows scaled by r(row) in fp32)
//   cache[row][head][t] <- k, v;   ctx[row][head] = softmax(q·K[0..t]^T + bias) · V[0..t]
// in ONE launch instead of two (QKV projection, then single-query attention): the projection -> attention seam is an
// all-to-all over features only ACROSS heads; within a head it is local.  Workgroup = (head, group of RG rows), 8 waves:
//   phase A  the 192 output features of the head (2 + 2 + 2 weight row tiles of 32) for the 32-row tile that holds the
//            group: waves split K (fixed slices, fixed-order LDS reduction -> deterministic); the MFMA computes all 32
//            rows of the tile, the workgroup keeps its RG columns (matrix-pipe time is not what bounds this kernel:
//            the per-CU ingest of the head's 192 x d weight slice is, and that is shared by the RG rows);
//   phase B  single-query attention of each row over its own cache stream + the position just produced (kept in LDS),
//            8 / RG waves per row (or RG / 8 rows per wave), online softmax, fixed-order merge.
// The first K / V round of phase B is issued before phase A: the cache stream does not depend on the projection.
// All workgroups of a head are 8 apart in the grid = on one XCD: the head's weight slice reaches that L2 once.
#include "mg_kernels.h"

namespace mg {

constexpr float QF_NEG = -1.0e30f;
constexpr int QF_NW = 8;

template <int RG>
__global__ __launch_bounds__(512) void qkv_attn_step_kernel(QkvStepArgs a) {
    MG_DYN_SMEM(smem);
    constexpr int NWR = RG >= QF_NW ? 1 : QF_NW / RG;       // waves per row in phase B
    constexpr int RPW = RG >= QF_NW ? RG / QF_NW : 1;       // rows per wave in phase B
    constexpr int UB = 2;                                   // K/V rounds (8 keys each) in flight per wave
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int sub = lane & 7, ks = lane >> 3;
    const int h = blockIdx.x % a.H, g = blockIdx.x / a.H;
    const int r0 = g * RG, rt = r0 >> 5, c0 = r0 & 31;
    const int inner = a.H * 64;
    const int tcur = a.t_dev ? *a.t_dev : a.t;
    // LDS: red[8][6][16][2*RG] f32 | rsl[RG] f32 | qkv[RG][192] bf16 | mrg[RG][NWR][8][10] f32
    float* red = (float*)smem;
    float* rsl = red + QF_NW * 6 * 16 * 2 * RG;
    uint16_t* qkv = (uint16_t*)(rsl + ((RG + 3) & ~3));
    float* mrg = (float*)(qkv + RG * 192);

    // ---- phase B prefetch: first K/V round of this wave's first row
    const int wi = RG >= QF_NW ? 0 : w / RG;                // index of this wave among its row's waves
    const int row_b0 = RG >= QF_NW ? w * RPW : w % RG;      // first (local) row of this wave
    auto kv_off = [&](int lrow, int kc) {
        const int row = r0 + lrow < a.rows ? r0 + lrow : a.rows - 1;
        const int prow = a.anc ? a.anc[(size_t)kc * a.rows + row] : row;
        return (((size_t)prow * a.H + h) * (size_t)a.cap + (size_t)kc) * 64 + sub * 8;
    };
    uint4 kn[UB], vn[UB];
    {
        const int kb = wi * 8 * UB;
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            int kc = kb + u * 8 + ks;
            kc = kc < tcur ? kc : (tcur > 0 ? tcur - 1 : 0);
            const size_t off = kv_off(row_b0, kc);
            kn[u] = tcur > 0 ? ld16(a.Kc + off) : make_uint4(0, 0, 0, 0);
            vn[u] = tcur > 0 ? ld16(a.Vc + off) : make_uint4(0, 0, 0, 0);
        }
    }
    // ---- phase A: projections
    // deferred RMSNorm scale of the group's rows: 8 threads per row sum the partials (fixed order)
    {
        const int lr = tid >> 3, j = tid & 7;
        float s = 0.f;
        if (a.rs.part && lr < RG) {
            const int row = r0 + lr < a.rows ? r0 + lr : a.rows - 1;
            const int per = a.rs.nparts >> 3;
            const float* p = a.rs.part + (size_t)row * a.rs.nparts + j * per;
            if ((per & 3) == 0) {
                for (int i = 0; i < per; i += 4) { const float4 v = *(const float4*)(p + i); s += (v.x + v.y) + (v.z + v.w); }
            } else {
                for (int i = 0; i < per; ++i) s += p[i];
            }
        }
        s = sum8(s);
        if (j == 0 && lr < RG) rsl[lr] = a.rs.part ? rsqrtf(s * a.rs.inv_d + a.rs.eps) : 1.0f;
    }
    const int kt16 = a.d >> 4;
    const int per = (kt16 + QF_NW - 1) / QF_NW;
    const int k0 = w * per, k1 = (k0 + per) < kt16 ? (k0 + per) : kt16;
    const int xkts = a.x_kts ? a.x_kts : kt16;
    const char* xp = (const char*)a.X + ((size_t)rt * xkts) * TILE_BYTES + lane * 16;
    const char* wp[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int tile = (j >> 1) * (inner >> 5) + 2 * h + (j & 1);
        wp[j] = (const char*)(a.W + pk_tile_off(tile, 0, a.d)) + lane * 16;
    }
    f32x16 acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[j] = acc_zero();
    constexpr int UA = 4;                                   // k-tiles per round: 24 weight + 4 activation fragments in flight
    for (int kt = k0; kt < k1; kt += UA) {
        uint4 wf[UA][6], xf[UA];
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int kk = kt + u < k1 ? kt + u : k1 - 1;    // past the slice: re-read, contribution masked below
#pragma unroll
            for (int j = 0; j < 6; ++j) wf[u][j] = ld16(wp[j] + (size_t)kk * TILE_BYTES);    // (re-read from L2 by the head's other groups)
            xf[u] = ld16(xp + (size_t)kk * TILE_BYTES);
        }
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            if (kt + u < k1) {
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[j] = mfma32(wf[u][j], xf[u], acc[j]);
            }
        }
    }
    // this lane's column = row c = l32 of the tile; keep columns c0 .. c0 + RG - 1
    if (l32 >= c0 && l32 < c0 + RG) {
        const int col = half * RG + (l32 - c0);
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((w * 6 + j) * 16 + r) * (2 * RG) + col] = acc[j][r];
    }
    __syncthreads();
    for (int o = tid; o < RG * 192; o += 512) {
        const int lr = o / 192, f = o - lr * 192;
        const int j = f >> 5, fr = f & 31;
        const int hf = (fr >> 2) & 1, r = (fr & 3) + 4 * (fr >> 3);
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < QF_NW; ++ww) s += red[((ww * 6 + j) * 16 + r) * (2 * RG) + hf * RG + lr];
        const uint16_t b = f32_to_bf16_rn(s * rsl[lr]);
        qkv[lr * 192 + f] = b;
        const int row = r0 + lr;
        if (f >= 64 && row < a.rows) {
            uint16_t* dst = (f < 128 ? a.Kc_w : a.Vc_w) + (((size_t)row * a.H + h) * (size_t)a.cap + (size_t)tcur) * 64 + (f & 63);
            *dst = b;
        }
    }
    __syncthreads();
    // ---- phase B: attention
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int lr = row_b0 + rr;
        const uint4 q = *(const uint4*)(qkv + lr * 192 + sub * 8);
        float m = QF_NEG, l = 0.f, o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (rr > 0) {      // later rows of this wave: their first round is issued here
            const int kb = wi * 8 * UB;
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                int kc = kb + u * 8 + ks;
                kc = kc < tcur ? kc : (tcur > 0 ? tcur - 1 : 0);
                const size_t off = kv_off(lr, kc);
                kn[u] = tcur > 0 ? ld16(a.Kc + off) : make_uint4(0, 0, 0, 0);
                vn[u] = tcur > 0 ? ld16(a.Vc + off) : make_uint4(0, 0, 0, 0);
            }
        }
        for (int kb = wi * 8 * UB; kb < tcur; kb += NWR * 8 * UB) {
            uint4 kv[UB], vv[UB];
            int key[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) { kv[u] = kn[u]; vv[u] = vn[u]; key[u] = kb + u * 8 + ks; }
            const int nb = kb + NWR * 8 * UB;
            if (nb < tcur) {
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    int kc = nb + u * 8 + ks;
                    kc = kc < tcur ? kc : tcur - 1;
                    const size_t off = kv_off(lr, kc);
                    kn[u] = ld16(a.Kc + off);
                    vn[u] = ld16(a.Vc + off);
                }
            }
            float s[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                float p = dot2_bf16(q.x, kv[u].x, 0.f);
                p = dot2_bf16(q.y, kv[u].y, p);
                p = dot2_bf16(q.z, kv[u].z, p);
                p = dot2_bf16(q.w, kv[u].w, p);
                p = sum8(p);
                int dist = tcur - key[u];
                dist = dist < 0 ? 0 : dist;
                s[u] = key[u] < tcur ? p + (a.bias ? a.bias[(size_t)dist * a.H + h] : 0.f) : QF_NEG;
            }
            float mx = s[0];
#pragma unroll
            for (int u = 1; u < UB; ++u) mx = fmaxf(mx, s[u]);
            const float mn = fmaxf(m, mx);
            const float al = fast_exp(m - mn);
            m = mn;
            l *= al;
#pragma unroll
            for (int dd = 0; dd < 8; ++dd) o8[dd] *= al;
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const float p = key[u] < tcur ? fast_exp(s[u] - mn) : 0.f;
                l += p;
                o8[0] += p * bf16lo(vv[u].x); o8[1] += p * bf16hi(vv[u].x);
                o8[2] += p * bf16lo(vv[u].y); o8[3] += p * bf16hi(vv[u].y);
                o8[4] += p * bf16lo(vv[u].z); o8[5] += p * bf16hi(vv[u].z);
                o8[6] += p * bf16lo(vv[u].w); o8[7] += p * bf16hi(vv[u].w);
            }
        }
        // the position just produced (distance 0): first wave of the row, key slot 0 (whole wave runs the shuffles)
        {
            const uint4 knew = *(const uint4*)(qkv + lr * 192 + 64 + sub * 8);
            const uint4 vnew = *(const uint4*)(qkv + lr * 192 + 128 + sub * 8);
            float p = dot2_bf16(q.x, knew.x, 0.f);
            p = dot2_bf16(q.y, knew.y, p);
            p = dot2_bf16(q.z, knew.z, p);
            p = dot2_bf16(q.w, knew.w, p);
            p = sum8(p);
            if (wi == 0 && ks == 0) {
                const float sc = p + (a.bias ? a.bias[h] : 0.f);
                const float mn = fmaxf(m, sc);
                const float al = fast_exp(m - mn), pe = fast_exp(sc - mn);
                m = mn;
                l = l * al + pe;
                o8[0] = o8[0] * al + pe * bf16lo(vnew.x); o8[1] = o8[1] * al + pe * bf16hi(vnew.x);
                o8[2] = o8[2] * al + pe * bf16lo(vnew.y); o8[3] = o8[3] * al + pe * bf16hi(vnew.y);
                o8[4] = o8[4] * al + pe * bf16lo(vnew.z); o8[5] = o8[5] * al + pe * bf16hi(vnew.z);
                o8[6] = o8[6] * al + pe * bf16lo(vnew.w); o8[7] = o8[7] * al + pe * bf16hi(vnew.w);
            }
        }
        // merge the 8 key slots of the wave (fixed tree)
#pragma unroll
        for (int step = 8; step <= 32; step <<= 1) {
            const float mo = __shfl_xor(m, step), lo = __shfl_xor(l, step);
            const float M = fmaxf(m, mo);
            const float f1 = fast_exp(m - M), f2 = fast_exp(mo - M);
            l = l * f1 + lo * f2;
#pragma unroll
            for (int dd = 0; dd < 8; ++dd) o8[dd] = o8[dd] * f1 + __shfl_xor(o8[dd], step) * f2;
            m = M;
        }
        if (ks == 0) {
            float* rp = mrg + ((size_t)(lr * NWR + wi) * 8 + sub) * 10;
            rp[0] = m; rp[1] = l;
#pragma unroll
            for (int dd = 0; dd < 8; ++dd) rp[2 + dd] = o8[dd];
        }
    }
    __syncthreads();
    // final merge of a row's NWR partials (fixed order) by 8 lanes; rows are dealt to the 64 lane-octets of the workgroup
    for (int lr = tid >> 3; lr < RG; lr += 64) {
        const int sb = tid & 7;
        float M = QF_NEG;
        for (int ww = 0; ww < NWR; ++ww) M = fmaxf(M, mrg[((size_t)(lr * NWR + ww) * 8 + sb) * 10]);
        float L = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int ww = 0; ww < NWR; ++ww) {
            const float* rp = mrg + ((size_t)(lr * NWR + ww) * 8 + sb) * 10;
            const float f = fast_exp(rp[0] - M);
            L += rp[1] * f;
#pragma unroll
            for (int dd = 0; dd < 8; ++dd) o[dd] += rp[2 + dd] * f;
        }
        const float inv = L > 0.f ? 1.0f / L : 0.f;
        const int row = r0 + lr;
        if (row < a.rows)
            st16(a.ctx + pk_off(row, a.ctx_col0 + h * 64 + sb * 8, a.ctx_ld ? a.ctx_ld : a.H * 64),
                 make_uint4(pack_bf16(o[0] * inv, o[1] * inv), pack_bf16(o[2] * inv, o[3] * inv),
                            pack_bf16(o[4] * inv, o[5] * inv), pack_bf16(o[6] * inv, o[7] * inv)));
    }
}

template <int RG>
static void launch_qkv_attn(const QkvStepArgs& a, mgStream_t stream) {
    constexpr int NWR = RG >= QF_NW ? 1 : QF_NW / RG;
    const int ngrp = (a.rows + RG - 1) / RG;
    const size_t sh = (size_t)QF_NW * 6 * 16 * 2 * RG * 4 + (size_t)((RG + 3) & ~3) * 4 + (size_t)RG * 192 * 2 + (size_t)RG * NWR * 8 * 10 * 4;
    static bool once = false;
    if (!once) { MG_SET_MAX_SMEM((&qkv_attn_step_kernel<RG>), sh); once = true; }
    MG_LAUNCH((qkv_attn_step_kernel<RG>), dim3(a.H * ngrp), dim3(512), sh, stream, a);
}

int qkv_attention_rows_per_group(int rows, int H) {
    int rg = 1;
    while (rg < 16 && (long)H * ((rows + rg - 1) / rg) > 320) rg <<= 1;     // about one workgroup per CU (16: LDS of phase A)
    return rg;
}

void qkv_attention_step(const QkvStepArgs& a, mgStream_t stream) {
    switch (a.rg) {
        case 1: launch_qkv_attn<1>(a, stream); break;
        case 2: launch_qkv_attn<2>(a, stream); break;
        case 4: launch_qkv_attn<4>(a, stream); break;
        case 8: launch_qkv_attn<8>(a, stream); break;
        default: launch_qkv_attn<16>(a, stream); break;
    }
}

}  // namespace mg
