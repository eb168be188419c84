// bf16 MFMA GEMMs of the VTL encoder / CXSMILES decoder (QKV, O, FFN wi/wo, cross-K/V, patch embed, lm_head).
//   out[m][n] = sum_k X[m][k] * W[n][k]      (nn.Linear: W is [out,in]; stock:412-415, 313-314)
// Both operands live in HBM in the packed fragment-tile format (mg_device.h), so
//   * a wave stages one 32x16 fragment with ONE global_load_lds (1 KiB contiguous in HBM, lane-linear in LDS),
//   * every ds_read_b128 of a fragment is base + 16*lane: conflict-free without swizzling,
//   * swapping the two MFMA operands transposes the accumulator for free, which lets each epilogue store
//     16-byte chunks in the layout its consumer wants (packed rows, packed transposed, or row-major fp32).
#include "k_gemm_epi.h"
#include <atomic>

namespace mg {

// ---------------------------------------------------------------------------------------------------------
// large-M GEMM: 128x128x64 block tile, 4 waves (2x2), wave tile 64x64 = 2x2 MFMA 32x32x16 accumulators,
// double-buffered LDS filled by global_load_lds (32 KiB per stage), one barrier per K-step.
// ---------------------------------------------------------------------------------------------------------
constexpr int GB_M = 128, GB_N = 128, GB_K = 64;
constexpr int GB_STAGE_BYTES = (GB_M + GB_N) / 32 * (GB_K / 16) * TILE_BYTES;   // 32 KiB

template <int EPI>
__global__ __launch_bounds__(256) void gemm_big_kernel(GemmArgs a) {
    MG_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nbn = (a.N + GB_N - 1) / GB_N;
    const int nbm = (a.M + GB_M - 1) / GB_M;
    int bid = blockIdx.x;
    const int nblk = nbm * nbn;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);   // XCD-contiguous tile ranges (speed only)
    const int bm = bid / nbn, bn = bid - bm * nbn;
    const int mt32 = (a.M + 31) >> 5, nt32 = (a.N + 31) >> 5, kt16 = a.K >> 4;
    const int nks = a.K / GB_K;

    // loader: wave w stages fragments f = 8w .. 8w+7 of the stage; f < 16: X row-tile f/4, k-tile f%4; else W.
    const char* src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = w * 8 + i, ff = f & 15, rt = ff >> 2, kt = ff & 3;
        const bool isW = f >= 16;
        int trow = isW ? (bn * 4 + rt) : (bm * 4 + rt);
        const int tmax = isW ? nt32 - 1 : mt32 - 1;
        trow = trow < tmax ? trow : tmax;   // clamp: tiles past the edge re-read the last tile, results are discarded
        const uint16_t* basep = isW ? a.W : a.X;
        src[i] = (const char*)(basep + pk_tile_off(trow, kt, a.K)) + lane * 16;
    }
    auto stage = [&](int buf, int ks) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            glds16(src[i] + (size_t)ks * (4 * TILE_BYTES), smem + buf * GB_STAGE_BYTES + (w * 8 + i) * TILE_BYTES);
    };

    const int wr = w >> 1, wc = w & 1;
    const int m0w = bm * GB_M + wr * 64, n0w = bn * GB_N + wc * 64;
    bool tor;
    if (EPI == EPI_F32_STORE || EPI == EPI_F32_RESID) tor = false;
    else if (EPI == EPI_HEADS) tor = !heads_region_is_T(a.heads, n0w < a.N ? n0w : 0);
    else tor = true;                                   // packed epilogues and EPI_RESID_NORM: lane owns a token

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = acc_zero();

    stage(0, 0);
    __syncthreads();
    for (int ks = 0; ks < nks; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nks) stage(cur ^ 1, ks + 1);
        const char* xb = smem + cur * GB_STAGE_BYTES + lane * 16;
        const char* wb = xb + 16 * TILE_BYTES;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            uint4 xf[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                xf[i] = ld16(xb + ((wr * 2 + i) * 4 + kt) * TILE_BYTES);
                wf[i] = ld16(wb + ((wc * 2 + i) * 4 + kt) * TILE_BYTES);
            }
            if (tor) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(wf[j], xf[i], acc[i][j]);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(xf[i], wf[j], acc[i][j]);
            }
        }
        __syncthreads();
    }

    if constexpr (EPI == EPI_RESID_NORM) {
#pragma unroll
        for (int i = 0; i < 2; ++i) resid_norm_epilogue(a, acc[i][0], acc[i][1], m0w + 32 * i, n0w, lane);
        return;
    }
    float rsv[2];
    { const int mrow2[2] = {m0w, m0w + 32}; row_scales_tiles<2>(a.rs, mrow2, a.M, lane, rsv); }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m0 = m0w + 32 * i, n0 = n0w + 32 * j;
            if constexpr (EPI == EPI_F32_STORE || EPI == EPI_F32_RESID) {
                tile_epilogue<EPI, false>(a, acc[i][j], m0, n0, lane);
            } else if constexpr (EPI == EPI_HEADS) {
                if (tor) tile_epilogue<EPI_HEADS, true, true>(a, acc[i][j], m0, n0, lane, 3, rsv[i]);
                else tile_epilogue<EPI_HEADS, false, true>(a, acc[i][j], m0, n0, lane, 3, rsv[i]);
            } else if constexpr (EPI != EPI_RESID_NORM) {
                tile_epilogue<EPI, true, true>(a, acc[i][j], m0, n0, lane, 3, rsv[i]);
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// large-M GEMM, wide form: 256x128x64 block tile, 8 waves (4x2, 64x64 per wave), THREE LDS stages of 48 KiB filled by
// global_load_lds with a prefetch distance of two K-steps.  One raw s_barrier per K-step and a COUNTED vmcnt: the
// copies of the next stage stay in flight across the barrier (PMC on the 2-stage kernel: 58 % of wave cycles parked
// at the barrier's vmcnt(0)).  Fragment-order LDS image => every ds_read_b128 is base + 16*lane, conflict-free.
// ---------------------------------------------------------------------------------------------------------
constexpr int GW_M = 256, GW_N = 128, GW_K = 64;
constexpr int GW_FRAGS = (GW_M + GW_N) / 32 * (GW_K / 16);      // 48 fragments per stage
constexpr int GW_STAGE_BYTES = GW_FRAGS * TILE_BYTES;           // 48 KiB
constexpr int GW_STAGES = 3;

template <int EPI>
__global__ __launch_bounds__(512) void gemm_wide_kernel(GemmArgs a) {
    MG_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nbn = (a.N + GW_N - 1) / GW_N;
    const int nbm = (a.M + GW_M - 1) / GW_M;
    int bid = blockIdx.x;
    const int nblk = nbm * nbn;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);   // XCD-contiguous tile ranges (speed only)
    const int bm = bid / nbn, bn = bid - bm * nbn;
    const int mt32 = (a.M + 31) >> 5, nt32 = (a.N + 31) >> 5;
    const int nks = a.K / GW_K;

    // loader: wave w copies fragments 6w .. 6w+5 of a stage; fragments 0..31 = X (row-tile f/4, k-tile f%4), 32..47 = W
    const char* src[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int f = w * 6 + i;
        const bool isW = f >= 32;
        const int ff = isW ? f - 32 : f, rt = ff >> 2, kt = ff & 3;
        int trow = isW ? (bn * 4 + rt) : (bm * 8 + rt);
        const int tmax = isW ? nt32 - 1 : mt32 - 1;
        trow = trow < tmax ? trow : tmax;
        src[i] = (const char*)((isW ? a.W : a.X) + pk_tile_off(trow, kt, a.K)) + lane * 16;
    }
    auto stage = [&](int buf, int ks) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            glds16_async(src[i] + (size_t)ks * (4 * TILE_BYTES), smem + buf * GW_STAGE_BYTES + (w * 6 + i) * TILE_BYTES);
    };

    const int wr = w >> 1, wc = w & 1;
    const int m0w = bm * GW_M + wr * 64, n0w = bn * GW_N + wc * 64;
    bool tor;
    if (EPI == EPI_F32_STORE || EPI == EPI_F32_RESID) tor = false;
    else if (EPI == EPI_HEADS) tor = !heads_region_is_T(a.heads, n0w < a.N ? n0w : 0);
    else tor = true;                                   // packed epilogues and EPI_RESID_NORM: lane owns a token

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = acc_zero();

    stage(0, 0);
    if (nks > 1) stage(1, 1);
    int cur = 0;
    for (int ks = 0; ks < nks; ++ks) {
        // own copies of stage ks have landed (those of stage ks+1 may still be in flight) ...
        if (ks + 1 < nks) MG_WAIT_VMCNT(6); else MG_WAIT_VMCNT(0);
        // ... and after the barrier everybody's have, and everybody is done reading the buffer refilled below
        MG_BARRIER_RAW();
        if (ks + 2 < nks) { int nb = cur + 2; nb = nb >= GW_STAGES ? nb - GW_STAGES : nb; stage(nb, ks + 2); }
        const char* xb = smem + cur * GW_STAGE_BYTES + lane * 16;
        const char* wb = xb + 32 * TILE_BYTES;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            uint4 xf[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                xf[i] = ld16(xb + ((wr * 2 + i) * 4 + kt) * TILE_BYTES);
                wf[i] = ld16(wb + ((wc * 2 + i) * 4 + kt) * TILE_BYTES);
            }
            if (tor) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(wf[j], xf[i], acc[i][j]);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(xf[i], wf[j], acc[i][j]);
            }
        }
        cur = cur + 1 >= GW_STAGES ? 0 : cur + 1;
    }

    if constexpr (EPI == EPI_RESID_NORM) {
#pragma unroll
        for (int i = 0; i < 2; ++i) resid_norm_epilogue(a, acc[i][0], acc[i][1], m0w + 32 * i, n0w, lane);
        return;
    }
    float rsv[2];
    { const int mrow2[2] = {m0w, m0w + 32}; row_scales_tiles<2>(a.rs, mrow2, a.M, lane, rsv); }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m0 = m0w + 32 * i, n0 = n0w + 32 * j;
            if constexpr (EPI == EPI_F32_STORE || EPI == EPI_F32_RESID) {
                tile_epilogue<EPI, false>(a, acc[i][j], m0, n0, lane);
            } else if constexpr (EPI == EPI_HEADS) {
                if (tor) tile_epilogue<EPI_HEADS, true, true>(a, acc[i][j], m0, n0, lane, 3, rsv[i]);
                else tile_epilogue<EPI_HEADS, false, true>(a, acc[i][j], m0, n0, lane, 3, rsv[i]);
            } else if constexpr (EPI != EPI_RESID_NORM) {
                tile_epilogue<EPI, true, true>(a, acc[i][j], m0, n0, lane, 3, rsv[i]);
            }
        }
}


// ---------------------------------------------------------------------------------------------------------
// large-M GEMM, 256x256x64 block tile, 8 waves as 2 (M) x 4 (N), 128x64 per wave (4x2 MFMA tiles, 128 accumulator
// registers).  Per K-step a wave issues 24 ds_read_b128 for 32 MFMAs (0.75 LDS reads per MFMA; the 256x128 kernel
// with 64x64 per wave needs 1.0 and is LDS-bandwidth-bound: 8 waves x 16 KiB per K-step = 1024 cycles of the CU's
// 128 B/clk LDS port against 1024 cycles of MFMA per SIMD).  Two LDS stages of 64 KiB; the copies of K-step ks+1 are
// issued right after the barrier of step ks and have a whole compute phase (~2048 MFMA cycles) to land.
// ---------------------------------------------------------------------------------------------------------

// TI = row tiles (of 32) per wave: TI = 4 -> 256x256 block tile; TI = 5 -> 320x256 (0.7 LDS reads per MFMA, 160
// accumulator registers, two 72 KiB stages), used where it makes the tile count a whole number of rounds over the 256 CUs
// (M = 40960, N = 1024: 128 x 4 = 512 tiles instead of 640).
// XP != 0: timing experiments only (MG_GEMM_EXP, tools/kbench.py encgemm; WRONG results): 1 = the fragments of k-tile 0 are
// reused for k-tiles 1..3 (LDS read traffic / 4), 2 = only the first two K-steps are staged (no global traffic afterwards), 4 = only one
// of the wave's 2*TI output tiles is stored (epilogue / 10); sums combine
template <int EPI, int TI, int XP = 0>
__global__ __launch_bounds__(512) void gemm_xl_kernel(GemmArgs a) {
    MG_DYN_SMEM(smem);
    constexpr int BM = 64 * TI, XT = 2 * TI;                   // X row tiles per block (2 wave rows x TI)
    constexpr int FRAGS = (XT + 8) * 4;                        // fragments per stage
    constexpr int STAGE_BYTES = FRAGS * TILE_BYTES;
    constexpr int PER_WAVE = (FRAGS + 7) / 8;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nbn = (a.N + GX_N - 1) / GX_N;
    // row-tile list (GemmArgs::row_tiles): the block's XT row tiles are entries bm*XT .. of the list instead of consecutive tiles;
    // the grid is sized for all rows, blocks beyond the list's end leave at once
    const int n_list = a.row_tiles ? *a.n_row_tiles : 0;
    const int M_run = a.row_tiles ? n_list * 32 : a.M;
    const int nbm = (M_run + BM - 1) / BM;
    int bid = blockIdx.x;
    const int nblk = nbm * nbn;
    if (bid >= nblk) return;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);   // XCD-contiguous tile ranges (speed only)
    const int bm = bid / nbn, bn = bid - bm * nbn;
    const int mt32 = (a.M + 31) >> 5, nt32 = (a.N + 31) >> 5;
    const int nks = a.K / GX_K;

    // loader: wave w copies fragments PER_WAVE*w ..; fragments 0 .. 4*XT-1 = X (row-tile f/4, k-tile f%4), then W
    const char* src[PER_WAVE];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        int f = w * PER_WAVE + i;
        f = f < FRAGS ? f : FRAGS - 1;                          // (FRAGS % 8 == 0 for TI = 4, 5: no clamping happens)
        const bool isW = f >= 4 * XT;
        const int ff = isW ? f - 4 * XT : f, rt = ff >> 2, kt = ff & 3;
        int trow = isW ? (bn * 8 + rt) : (bm * XT + rt);
        if (!isW && a.row_tiles) trow = a.row_tiles[trow < n_list ? trow : n_list - 1];      // (entries past the end re-read the last live tile)
        const int tmax = isW ? nt32 - 1 : mt32 - 1;
        trow = trow < tmax ? trow : tmax;
        src[i] = (const char*)((isW ? a.W : a.X) + pk_tile_off(trow, kt, a.K)) + lane * 16;
    }
    const mg_lds_t sm0 = mg_lds_addr(smem);
    auto stage = [&](int buf, int ks) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i)
            if (w * PER_WAVE + i < FRAGS)
                glds16_async_lds(src[i] + (size_t)ks * (4 * TILE_BYTES), sm0 + buf * STAGE_BYTES + (w * PER_WAVE + i) * TILE_BYTES);
    };

    const int wr = w >> 2, wc = w & 3;
    const int m0w = bm * BM + wr * (32 * TI), n0w = bn * GX_N + wc * 64;
    bool tor;
    if (EPI == EPI_F32_STORE || EPI == EPI_F32_RESID) tor = false;
    else if (EPI == EPI_HEADS) tor = !heads_region_is_T(a.heads, n0w < a.N ? n0w : 0);
    else tor = true;                                   // packed epilogues and EPI_RESID_NORM: lane owns a token

    f32x16 acc[TI][2];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = acc_zero();

    // Fragment registers are double-buffered: the reads of k-tile kt+1 are issued BEFORE the MFMAs of k-tile kt, and the step's
    // barrier sits in front of the LAST k-tile's MFMAs, whose operands are in registers by then - so the next step's copies and
    // first fragment reads are issued under 10 MFMAs per wave instead of in front of an idle matrix pipe.  (The schedule hipcc
    // derives from a plain read-then-multiply loop overlapped one MFMA with the next k-tile's reads: measured 1.5 PFLOP/s with
    // all memory traffic and the epilogue switched off.)
    struct Frags { mg_raw16 x[TI], w[2]; };
    // per-lane LDS addresses of the wave's first X / W fragment in stage 0 (stage 1: + STAGE_BYTES); the fragment and k-tile
    // select an instruction immediate
    const mg_lds_t lx0 = sm0 + lane * 16 + wr * (TI * 4 * TILE_BYTES);
    const mg_lds_t lw0 = sm0 + lane * 16 + (4 * XT + wc * 8) * TILE_BYTES;
    auto rd = [&](int buf, auto ktc, Frags& f) {
        constexpr int kt = (XP & 1) ? 0 : decltype(ktc)::value;
        const mg_lds_t lx = lx0 + buf * STAGE_BYTES, lw = lw0 + buf * STAGE_BYTES;
        lds_rd16_async<(0 * 4 + kt) * TILE_BYTES>(f.w[0], lw);
        lds_rd16_async<(1 * 4 + kt) * TILE_BYTES>(f.w[1], lw);
        lds_rd16_async<(0 * 4 + kt) * TILE_BYTES>(f.x[0], lx);
        lds_rd16_async<(1 * 4 + kt) * TILE_BYTES>(f.x[1], lx);
        lds_rd16_async<(2 * 4 + kt) * TILE_BYTES>(f.x[2], lx);
        lds_rd16_async<(3 * 4 + kt) * TILE_BYTES>(f.x[3], lx);
        if constexpr (TI > 4) lds_rd16_async<(4 * 4 + kt) * TILE_BYTES>(f.x[TI - 1], lx);
    };
    // all outstanding fragment reads of the wave have landed; the MFMAs below cannot be moved above this point
    auto landed = [&](Frags& f) {
        MG_WAIT_LGKM_TIE(0, f.w[0]);
        MG_TIE(f.w[1]);
#pragma unroll
        for (int i = 0; i < TI; ++i) MG_TIE(f.x[i]);
    };
    auto mm = [&](const Frags& f) {
        uint4 xw[2], xx[TI];
#pragma unroll
        for (int j = 0; j < 2; ++j) xw[j] = raw16_get(f.w[j]);
#pragma unroll
        for (int i = 0; i < TI; ++i) xx[i] = raw16_get(f.x[i]);
        if (tor) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(xw[j], xx[i], acc[i][j]);
        } else {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(xx[i], xw[j], acc[i][j]);
        }
    };
    Frags fa, fb;
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
    stage(0, 0);
    MG_WAIT_VMCNT(0);
    MG_BARRIER_RAW();
    if (nks > 1) stage(1, 1);
    rd(0, K0{}, fa);
    for (int ks = 0; ks < nks; ++ks) {
        const int cur = ks & 1;
        landed(fa);
        rd(cur, K1{}, fb);
        MG_SCHED_FENCE();            // (the reads stay in front of the MFMAs they are to be hidden under)
        mm(fa);
        MG_SCHED_FENCE();
        landed(fb);
        rd(cur, K2{}, fa);
        MG_SCHED_FENCE();
        mm(fb);
        MG_SCHED_FENCE();
        landed(fa);
        rd(cur, K3{}, fb);
        MG_SCHED_FENCE();
        mm(fa);
        MG_SCHED_FENCE();
        landed(fb);                  // (also: own reads of this buffer are complete ...
        if (ks + 1 < nks) {
            MG_WAIT_VMCNT(0);        // ... own copies of the next step (issued one step ago) have landed ...
            MG_BARRIER_RAW();        // ... and so have everybody's)
            if (ks + 2 < nks && !(XP & 2)) stage(cur, ks + 2);
            rd(cur ^ 1, K0{}, fa);
        }
        MG_SCHED_FENCE();
        mm(fb);
        MG_SCHED_FENCE();
    }

    int mrow[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        if (a.row_tiles) {
            const int e = bm * XT + wr * TI + i;
            mrow[i] = e < n_list ? a.row_tiles[e] * 32 : a.M;        // past the list's end: row index M, every store is guarded by m < M
        } else {
            mrow[i] = m0w + 32 * i;
        }
    }
    xl_epilogue<EPI, TI, XP>(a, acc, mrow, n0w, tor, lane);
}
template <int EPI, int TI>
static void launch_xl(const GemmArgs& a, mgStream_t stream) {
    constexpr int BM = 64 * TI;
    const int nblk = ((a.M + BM - 1) / BM) * ((a.N + GX_N - 1) / GX_N);
    const size_t sh = (size_t)2 * (2 * TI + 8) * 4 * TILE_BYTES;
    static bool once = false;
    if (!once) { MG_SET_MAX_SMEM((&gemm_xl_kernel<EPI, TI>), sh); once = true; }
#ifdef MG_TOOLS      // what-if variants with WRONG results: tools builds only (build.py tools), never in the product library
    if constexpr (EPI == EPI_PK && TI == 5) {          // timing experiments (see the kernel's XP parameter)
        static int xp = -1;
        if (xp < 0) { const char* e = getenv("MG_GEMM_EXP"); xp = e ? atoi(e) : 0; }
        if (xp) {
#define MG_XP(N) case N: { static bool o = false; if (!o) { MG_SET_MAX_SMEM((&gemm_xl_kernel<EPI, TI, N>), sh); o = true; } \
                             MG_LAUNCH((gemm_xl_kernel<EPI, TI, N>), dim3(nblk), dim3(512), sh, stream, a); } break;
            switch (xp) { MG_XP(1) MG_XP(2) MG_XP(3) MG_XP(4) MG_XP(5) MG_XP(6) default: MG_XP(7) }
#undef MG_XP
            return;
        }
    }
#endif
    MG_LAUNCH((gemm_xl_kernel<EPI, TI>), dim3(nblk), dim3(512), sh, stream, a);
}

template <int EPI>
static void launch_wide(const GemmArgs& a, mgStream_t stream) {
    const int nblk = ((a.M + GW_M - 1) / GW_M) * ((a.N + GW_N - 1) / GW_N);
    const size_t sh = (size_t)GW_STAGES * GW_STAGE_BYTES;
    static bool once = false;
    if (!once) { MG_SET_MAX_SMEM(&gemm_wide_kernel<EPI>, sh); once = true; }
    MG_LAUNCH((gemm_wide_kernel<EPI>), dim3(nblk), dim3(512), sh, stream, a);
}

// 0: 128x128 two-stage kernel only; 1: + 256x128 three-stage kernel for M >= 256; 2: + 256x256 kernel wherever it fits;
// 4: + 320x256; 5 / 6: persistent ping-pong kernel with 256- / 320-row tiles (k_gemm_pp.hip);
// 3 (default): by shape = 6 where the ping-pong kernel applies, else 4
// (the three test / A-B switches of this file are process-wide: atomics, so that a host thread of another execution context never reads a torn
// value; they select among kernels with IDENTICAL results and are meant to be set while no call is running - see include/mgrapher.h)
static std::atomic<int> g_gemm_variant{3};
void gemm_set_variant(int v) { g_gemm_variant = v; }

bool gemm_has_gelu_epilogue(int M, int N) { return g_gemm_variant >= 2 && M >= 320 && N >= GX_N; }

void gemm(const GemmArgs& a, int epi, mgStream_t stream) {
    static bool env_read = false;
    if (!env_read) { env_read = true; if (const char* e = getenv("MG_GEMM_VARIANT")) g_gemm_variant = atoi(e); }   // A/B runs
    const int gv = g_gemm_variant;          // one read per call (the switch is process-wide)
    // default (3): the persistent ping-pong kernel with 320-row tiles where its shape rules hold (K % 128 == 0), measured against the
    // two-stage kernel at M = 40960 (us): QKV 292 -> 245, O 166 -> 156, wi 356 -> 301, wo 434 -> 396, cross-KV 154 -> 146 (profiles/r04_b_*)
    // Small problems (round 5: the ChemicalOCR prefill at M = 4096, N = 576; the last Swin stages): fewer 320 x 256 tiles than
    // CUs leave part of the chip idle (o_proj of the prefill: 32 workgroups, 80 us for 2.7 GFLOP) - they take the 256 x 128 / 128 x 128
    // kernels, whose grids are 2.5 - 5 x larger.  Every output element is the same k-ascending chain of MFMAs in all tile kernels.
    static int small_env = -1;
    if (small_env < 0) { const char* e = getenv("MG_GEMM_SMALL"); small_env = e ? atoi(e) : 256; if (small_env == 1) small_env = 256; }      // (the tile-count threshold: fewer 320 x 256 tiles than CUs; 0: off.  96 -> 256: OCSR branch at 32 images 10.55 -> 9.68 ms, OCR tower + prefill 17.99 -> 17.84 ms)
    const long tiles_xl = (long)((a.M + 319) / 320) * ((a.N + GX_N - 1) / GX_N);
    // (not for the main encoder's deferred-RMSNorm chain - gain / partial sums / row scales: its GEMMs of one geometry stay on ONE kernel family,
    //  with or without the live-row-tile list, so that the list changes no bits)
    const bool small = gv == 3 && small_env && tiles_xl < small_env && epi != EPI_PK_GELU && !a.row_tiles && !a.part && !a.gain && !a.rs.part;
    if (!small && (gv == 3 || gv == 5 || gv == 6) && a.M >= 320 && a.N >= GX_N) {        // ping-pong persistent kernel, TI = 4 (variant 5) / 5
        if (gemm_pp(a, epi, gv == 5 ? 4 : 5, stream)) return;
    }
    if (!small && gv >= 2 && a.M >= 320 && a.N >= GX_N) {
        // measured at M = 40960 (PFLOP/s, 256x128 / 256x256 / 320x256): QKV 0.72 / 0.92 / 1.01, O 0.47 / 0.45 / 0.53,
        // wi 0.78 / 0.99 / 1.04, wo 0.73 / 0.69 / 0.80, cross-KV 0.97 / 1.10 / 1.09 -> the 320-row tile by default
        const bool five = gv >= 3;
        const bool four = gv == 2;
        if (five) {
            switch (epi) {
                case EPI_F32_STORE: launch_xl<EPI_F32_STORE, 5>(a, stream); break;
                case EPI_F32_RESID: launch_xl<EPI_F32_RESID, 5>(a, stream); break;
                case EPI_PK_RELU: launch_xl<EPI_PK_RELU, 5>(a, stream); break;
                case EPI_PK_GELU: launch_xl<EPI_PK_GELU, 5>(a, stream); break;
                case EPI_PK: launch_xl<EPI_PK, 5>(a, stream); break;
                case EPI_PK_BIAS: launch_xl<EPI_PK_BIAS, 5>(a, stream); break;
                case EPI_PK_GELU_ERF: launch_xl<EPI_PK_GELU_ERF, 5>(a, stream); break;
                case EPI_RESID_NORM: launch_xl<EPI_RESID_NORM, 5>(a, stream); break;
                default: launch_xl<EPI_HEADS, 5>(a, stream); break;
            }
            return;
        }
        if (four) {
            switch (epi) {
                case EPI_F32_STORE: launch_xl<EPI_F32_STORE, 4>(a, stream); break;
                case EPI_F32_RESID: launch_xl<EPI_F32_RESID, 4>(a, stream); break;
                case EPI_PK_RELU: launch_xl<EPI_PK_RELU, 4>(a, stream); break;
                case EPI_PK_GELU: launch_xl<EPI_PK_GELU, 4>(a, stream); break;
                case EPI_PK: launch_xl<EPI_PK, 4>(a, stream); break;
                case EPI_PK_BIAS: launch_xl<EPI_PK_BIAS, 4>(a, stream); break;
                case EPI_PK_GELU_ERF: launch_xl<EPI_PK_GELU_ERF, 4>(a, stream); break;
                case EPI_RESID_NORM: launch_xl<EPI_RESID_NORM, 4>(a, stream); break;
                default: launch_xl<EPI_HEADS, 4>(a, stream); break;
            }
            return;
        }
    }
    const long tiles_w = (long)((a.M + GW_M - 1) / GW_M) * ((a.N + GW_N - 1) / GW_N);
    if (gv >= 1 && a.M >= GW_M && !(small && tiles_w < 96)) {      // (fewer than 96 of the 256 x 128 tiles as well: the 128 x 128 kernel)
        switch (epi) {
            case EPI_F32_STORE: launch_wide<EPI_F32_STORE>(a, stream); break;
            case EPI_F32_RESID: launch_wide<EPI_F32_RESID>(a, stream); break;
            case EPI_PK_RELU: launch_wide<EPI_PK_RELU>(a, stream); break;
            case EPI_PK: launch_wide<EPI_PK>(a, stream); break;
            case EPI_PK_BIAS: launch_wide<EPI_PK_BIAS>(a, stream); break;
            case EPI_PK_GELU_ERF: launch_wide<EPI_PK_GELU_ERF>(a, stream); break;
            case EPI_RESID_NORM: launch_wide<EPI_RESID_NORM>(a, stream); break;
            default: launch_wide<EPI_HEADS>(a, stream); break;
        }
        return;
    }
    const int nblk = ((a.M + GB_M - 1) / GB_M) * ((a.N + GB_N - 1) / GB_N);
    const dim3 grid(nblk), block(256);
    const size_t sh = 2 * GB_STAGE_BYTES;
    switch (epi) {
        case EPI_F32_STORE: MG_LAUNCH((gemm_big_kernel<EPI_F32_STORE>), grid, block, sh, stream, a); break;
        case EPI_F32_RESID: MG_LAUNCH((gemm_big_kernel<EPI_F32_RESID>), grid, block, sh, stream, a); break;
        case EPI_PK_RELU: MG_LAUNCH((gemm_big_kernel<EPI_PK_RELU>), grid, block, sh, stream, a); break;
        case EPI_PK: MG_LAUNCH((gemm_big_kernel<EPI_PK>), grid, block, sh, stream, a); break;
        case EPI_PK_BIAS: MG_LAUNCH((gemm_big_kernel<EPI_PK_BIAS>), grid, block, sh, stream, a); break;
        case EPI_PK_GELU_ERF: MG_LAUNCH((gemm_big_kernel<EPI_PK_GELU_ERF>), grid, block, sh, stream, a); break;
        case EPI_RESID_NORM: MG_LAUNCH((gemm_big_kernel<EPI_RESID_NORM>), grid, block, sh, stream, a); break;
        default: MG_LAUNCH((gemm_big_kernel<EPI_HEADS>), grid, block, sh, stream, a); break;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Row-tile split of the decode-step projections.  With more than one 32-row tile of live sequences (beam search: 160 rows,
// the OCR stage at 128 pages) a workgroup used to walk all MT tiles itself - the kernels are latency-sized, so their time grew
// with MT (FFN-wo 6.1 us at 32 rows, 24.9 us at 160).  Instead the grid gets a second dimension: blockIdx.y = row tile, every
// workgroup runs the one-tile form on a shifted view of the arguments (the weight slice is re-read from L2 by the other tiles'
// workgroups, its first touch comes from HBM once).  Same arithmetic per row, same summation order: results are bit-identical.
// ---------------------------------------------------------------------------------------------------------
// One tile per workgroup only where the weights are small enough to stay in the XCDs' L2 while the row tiles' workgroups come and go (<= 4 MiB: the OCR text
// model's projections).  Measured (profiles/r02_rows_split_ab.txt): OCR stage at 128 pages 141 -> 163 pages/s; the main decoder's
// 6-18 MB projections at 160 beam rows got SLOWER when split (61.7 -> 51.2 images/s: every row tile's workgroups pull the weight
// slice from HBM again), so they keep walking their tiles with the weights in registers.
// Larger weights (the main decoder at beam search, 160 rows = 5 tiles, 6-18 MB per projection) do NOT split: one tile per workgroup
// measured 61.7 -> 51.2 images/s and two groups of tiles 67.4 -> 60.5 (profiles/r02_rows_split_ab.txt) - the other groups' workgroups
// pull the weight slice from HBM again instead of finding it in L2.
static std::atomic<int> g_rows_split_mode{-1};      // -1: by weight size (default); 0: never; 1: always one tile per workgroup  (tests, A/B runs)
void gemm_rows_set_split(int mode) { g_rows_split_mode = mode; }
static int rows_split_tiles(int mt, size_t weight_elems) {          // row tiles per workgroup; mt = no split
    static bool env_read = false;
    if (!env_read) { env_read = true; if (const char* e = getenv("MG_ROWS_SPLIT")) g_rows_split_mode = atoi(e); }
    if (mt <= 1 || g_rows_split_mode == 0) return mt;
    const bool small = weight_elems * sizeof(uint16_t) <= ((size_t)4 << 20);
    return (g_rows_split_mode == 1 || (g_rows_split_mode < 0 && small)) ? 1 : mt;
}
template <int EPI>
MG_DEV void shift_rows(GemmArgs& a, int rt) {
    if (rt == 0) return;
    const int xkts = a.x_kts ? a.x_kts : (a.K >> 4);
    a.X += (size_t)rt * xkts * TILE_ELEMS;
    if (a.out_f32) a.out_f32 += (size_t)rt * 32 * a.ldo;
    if (a.out_pk) a.out_pk += (size_t)rt * ((EPI == EPI_PK_SWIGLU ? (a.out_ld ? a.out_ld : a.N >> 1) : a.N) >> 4) * TILE_ELEMS;
    if constexpr (EPI == EPI_HEADS) {
#pragma unroll
        for (int ri = 0; ri < 3; ++ri) {
            if (!a.heads.ptr[ri]) continue;
            if (a.heads.fmt[ri] == HF_STEP_Q) a.heads.ptr[ri] += (size_t)rt * 32 * a.heads.H * 64;
            else if (a.heads.fmt[ri] == HF_STEP_KV && !a.heads.row_map) a.heads.ptr[ri] += (size_t)rt * 32 * a.heads.H * (size_t)a.heads.S_cap * 64;
        }
        if (a.heads.row_map) a.heads.row_map += 32 * rt;
    }
    if (a.rs.part) a.rs.part += (size_t)rt * 32 * a.rs.nparts;
    a.M -= 32 * rt;
}
MG_DEV void shift_rows(ResidArgs& a, int rt) {
    if (rt == 0) return;
    const int xkts = a.x_kts ? a.x_kts : (a.K >> 4);
    a.X += (size_t)rt * xkts * TILE_ELEMS;
    a.h += (size_t)rt * 32 * a.N;
    if (a.x_pk) a.x_pk += (size_t)rt * ((a.x_ld ? a.x_ld : a.N) >> 4) * TILE_ELEMS;
    if (a.x2_pk) a.x2_pk += (size_t)rt * ((a.x2_ld ? a.x2_ld : a.N) >> 4) * TILE_ELEMS;
    if (a.part) a.part += (size_t)rt * 32 * (a.N >> 3);
    if (a.rs.part) a.rs.part += (size_t)rt * 32 * a.rs.nparts;
    a.M -= 32 * rt;
}

// ---------------------------------------------------------------------------------------------------------
// decode-step GEMM (M <= 32*MT live sequences): HBM-bound weight streaming.  One workgroup per 32 output
// features; its 4 waves split K, each streaming its weight fragments straight to registers (one contiguous
// 1 KiB wave-load per fragment, no LDS round trip for a stream that is read once), the activation fragments
// come from L2.  Partial accumulators are combined through LDS in a fixed order (deterministic).
// ---------------------------------------------------------------------------------------------------------
// HALF: one workgroup per 16 output features (half a weight tile; the other half's lanes feed zeros to the MFMA) —
// doubles the number of workgroups for projections whose epilogue must see complete sums (relu, bf16 per-head stores).
template <int EPI, int MT, bool HALF, int NW, int U>
MG_DEV void rows_block(const GemmArgs& a, int bid, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nt = HALF ? (bid >> 1) : bid;
    const int sub = HALF ? (bid & 1) : 0;
    const bool wvalid = !HALF || (((lane & 31) >> 4) == sub);
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    const int kt16 = a.K >> 4;
    const int per = (kt16 + NW - 1) / NW;
    const int k0 = w * per, k1 = (k0 + per) < kt16 ? (k0 + per) : kt16;
    constexpr bool TOR = !(EPI == EPI_F32_STORE || EPI == EPI_F32_RESID);
    float* rsl = (float*)(smem + NW * 16 * 64 * sizeof(float));     // [32*MT] deferred RMSNorm scale per row
    const char* wp = (const char*)(a.W + pk_tile_off(nt, 0, a.K)) + lane * 16;
    const int xkts = a.x_kts ? a.x_kts : kt16;
    const char* xp = (const char*)a.X + ((size_t)a.x_k0 * TILE_BYTES + lane * 16);
    // the first round of the weight stream (HBM) is issued before anything else, so that the L2 round trip of the row
    // scales overlaps it instead of preceding it
    int kt = k0;
    uint4 wf[U];
    const bool first_full = kt + U <= k1;
    if (first_full) {
#pragma unroll
        for (int u = 0; u < U; ++u) wf[u] = wvalid ? ld16_stream(wp + (size_t)(kt + u) * TILE_BYTES) : zero4;
    }
    if (TOR) block_row_scales(a.rs, a.M, 32 * MT, rsl, tid, NW * 64);

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = acc_zero();
    for (bool first = true; kt + U <= k1; kt += U, first = false) {
        if (!first) {
#pragma unroll
            for (int u = 0; u < U; ++u) wf[u] = wvalid ? ld16_stream(wp + (size_t)(kt + u) * TILE_BYTES) : zero4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const uint4 xf = ld16(xp + ((size_t)i * xkts + (kt + u)) * TILE_BYTES);
                acc[i] = TOR ? mfma32(wf[u], xf, acc[i]) : mfma32(xf, wf[u], acc[i]);
            }
        }
    }
    for (; kt < k1; ++kt) {
        const uint4 wf = wvalid ? ld16_stream(wp + (size_t)kt * TILE_BYTES) : zero4;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const uint4 xf = ld16(xp + ((size_t)i * xkts + kt) * TILE_BYTES);
            acc[i] = TOR ? mfma32(wf, xf, acc[i]) : mfma32(xf, wf, acc[i]);
        }
    }
    // combine the NW K-slices through one LDS slab, one m-tile at a time: slab[w][r][lane]
    float* slab = (float*)smem;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[(w * 16 + r) * 64 + lane] = acc[i][r];
        __syncthreads();
        if constexpr (TOR && !HALF) {
            // whole tile, token-major epilogue: the two 16-feature halves (registers 0-7 / 8-15, one 16-byte chunk per
            // lane each) are reduced and stored by two waves in parallel
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (((2 * i + q) % NW) == w) {
                    f32x16 s = acc_zero();
#pragma unroll
                    for (int r = 8 * q; r < 8 * q + 8; ++r) {
                        float v = 0.f;
#pragma unroll
                        for (int ww = 0; ww < NW; ++ww) v += slab[(ww * 16 + r) * 64 + lane];
                        s[r] = v * rsl[32 * i + (lane & 31)];
                    }
                    tile_epilogue<EPI, TOR>(a, s, 32 * i, 32 * nt, lane, 1 << q);
                }
            }
        } else if ((i % NW) == w) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) v += slab[(ww * 16 + r) * 64 + lane];
                s[r] = TOR ? v * rsl[32 * i + (lane & 31)] : v;
            }
            tile_epilogue<EPI, TOR>(a, s, 32 * i, 32 * nt, lane, HALF ? (1 << sub) : 3);
        }
        __syncthreads();
    }
}
// Half-tile projections (16 output features per workgroup) on v_mfma_f32_16x16x32_bf16: no zero-padded feature rows, one
// MFMA per two k-tiles and token group, every lane of a weight wave-load carries data.  Epilogues: packed bf16 rows
// (optionally relu) or per-head stores; a lane holds 4 consecutive features of one token, two lanes (l, l^16) make one
// 16-byte chunk.  Operand addressing as in resid_block16.
// FT = 2 (several row tiles): the workgroup takes BOTH 16-feature halves of its weight tile - every activation fragment it pulls from L2
// feeds two MFMAs instead of one (at 5 row tiles a workgroup reads 10 KB of activations per KB of weights; the activation re-reads of
// all workgroups are what these launches cost beside other contexts).  Same K partition over the waves, same reduction order: the
// results are bit-identical to the FT = 1 form (tests/test_kernels.py).
template <int EPI, int MT, int NW, int U, int FT = 1>
MG_DEV void rows_block16(const GemmArgs& a, int bid, char* smem) {
    static_assert(EPI == EPI_PK || EPI == EPI_PK_RELU || EPI == EPI_HEADS || EPI == EPI_PK_SWIGLU || EPI == EPI_F32_STORE,
                  "half-tile form: packed, per-head, SwiGLU and plain fp32 epilogues");
    static_assert(FT == 1 || MT > 1, "both halves per workgroup: several row tiles only");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int nt = FT == 2 ? bid : bid >> 1, sub0 = FT == 2 ? 0 : (bid & 1);
    const int kt16 = a.K >> 4, kp = kt16 >> 1;
    const int per = (kp + NW - 1) / NW;
    const int p0 = w * per, p1 = (p0 + per) < kp ? (p0 + per) : kp;
    float* rsl = (float*)(smem + NW * 8 * FT * 64 * sizeof(float));     // [32*MT] deferred RMSNorm scale per row
    const size_t lane_off = (size_t)(kg >> 1) * TILE_BYTES + (size_t)(kg & 1) * 512;
    const char* wp = (const char*)(a.W + pk_tile_off(nt, 0, a.K)) + lane_off + (size_t)(16 * sub0 + r16) * 16;      // second half: + 256 B
    const int xkts = a.x_kts ? a.x_kts : kt16;
    const char* xp = (const char*)a.X + (size_t)a.x_k0 * TILE_BYTES + lane_off + (size_t)r16 * 16;
    RsRegs rsr;                                                  // load order = wait order, see resid_block16
    rs_issue(a.rs, a.M, 32 * MT, tid, NW * 64, rsr);
    int p = p0;
    uint4 wf[U][FT];
    constexpr bool XPF = MT == 1;                                // see resid_block16
    uint4 xf[XPF ? U : 1][2];
    if (p + U <= p1) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int f = 0; f < FT; ++f) wf[u][f] = ld16_stream(wp + (size_t)(p + u) * (2 * TILE_BYTES) + f * 256);
        if constexpr (XPF) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const char* xt = xp + (size_t)(2 * (p + u)) * TILE_BYTES;
                xf[u][0] = ld16(xt); xf[u][1] = ld16(xt + 256);
            }
        }
    }
    rs_finish(a.rs, a.M, 32 * MT, rsl, tid, NW * 64, rsr);
    f32x4 acc[MT][2][FT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int f = 0; f < FT; ++f) { acc[i][0][f] = acc4_zero(); acc[i][1][f] = acc4_zero(); }
    for (bool first = true; p + U <= p1; p += U, first = false) {
        if (!first) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int f = 0; f < FT; ++f) wf[u][f] = ld16_stream(wp + (size_t)(p + u) * (2 * TILE_BYTES) + f * 256);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const char* xt = xp + ((size_t)i * xkts + 2 * (p + u)) * TILE_BYTES;
                if (XPF && first) {
                    acc[i][0][0] = mfma16(wf[u][0], xf[XPF ? u : 0][0], acc[i][0][0]);
                    acc[i][1][0] = mfma16(wf[u][0], xf[XPF ? u : 0][1], acc[i][1][0]);
                } else {
                    const uint4 xa = ld16(xt), xb = ld16(xt + 256);
#pragma unroll
                    for (int f = 0; f < FT; ++f) {
                        acc[i][0][f] = mfma16(wf[u][f], xa, acc[i][0][f]);
                        acc[i][1][f] = mfma16(wf[u][f], xb, acc[i][1][f]);
                    }
                }
            }
        }
    }
    for (; p < p1; ++p) {
        uint4 w1[FT];
#pragma unroll
        for (int f = 0; f < FT; ++f) w1[f] = ld16_stream(wp + (size_t)p * (2 * TILE_BYTES) + f * 256);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const char* xt = xp + ((size_t)i * xkts + 2 * p) * TILE_BYTES;
            const uint4 xa = ld16(xt), xb = ld16(xt + 256);
#pragma unroll
            for (int f = 0; f < FT; ++f) {
                acc[i][0][f] = mfma16(w1[f], xa, acc[i][0][f]);
                acc[i][1][f] = mfma16(w1[f], xb, acc[i][1][f]);
            }
        }
    }
    float* slab = (float*)smem;                                  // [NW][8 * FT][64]
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) slab[(w * 8 * FT + f * 8 + g * 4 + j) * 64 + lane] = acc[i][g][f][j];
        __syncthreads();
        // unit (m-tile i, token group g, feature half f) is finished by wave ((2 i + g) FT + f) % NW: the waves share a tile's epilogue
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
          for (int f = 0; f < FT; ++f) {
            if ((((2 * i + g) * FT + f) % NW) == w) {
                const int sub = sub0 + f;
                const int m = 32 * i + 16 * g + r16;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = 0.f;
                    for (int ww = 0; ww < NW; ++ww) t += slab[(ww * 8 * FT + f * 8 + g * 4 + j) * 64 + lane];
                    t *= rsl[32 * i + 16 * g + r16];
                    v[j] = (EPI == EPI_PK_RELU) ? fmaxf(t, 0.f) : t;
                }
                if constexpr (EPI == EPI_F32_STORE) {
                    // 4 consecutive features of token m: one 16-byte store (the row scale was applied above; no bias slot here)
                    const int n = nt * 32 + 16 * sub + 4 * kg;
                    if (m < a.M && n < a.N) *(float4*)(a.out_f32 + (size_t)m * a.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
                    continue;
                }
                if constexpr (EPI == EPI_PK_SWIGLU) {
                    // (gate, up, gate, up) of two MLP features here, the partner lane holds the next two: 8 bytes of the output row
                    const uint32_t mine = pack_bf16(v[0] / (1.0f + fast_exp(-v[0])) * v[1], v[2] / (1.0f + fast_exp(-v[2])) * v[3]);
                    const uint32_t other = __shfl_xor(mine, 16);
                    const int n = nt * 32 + 16 * sub + 4 * kg;
                    if ((kg & 1) == 0 && m < a.M && n < a.N)
                        *(uint2*)(a.out_pk + pk_off(m, a.out_col0 + (n >> 1), a.out_ld ? a.out_ld : a.N >> 1)) = make_uint2(mine, other);
                    continue;
                }
                // features 4*kg .. 4*kg+3 of token m here; lanes with even kg collect the partner's four (kg + 1)
                const uint32_t lo = pack_bf16(v[0], v[1]), hi = pack_bf16(v[2], v[3]);
                const uint32_t plo = __shfl_xor(lo, 16), phi = __shfl_xor(hi, 16);
                if ((kg & 1) == 0 && m < a.M) {
                    const uint4 ch = make_uint4(lo, hi, plo, phi);
                    const int n = nt * 32 + 16 * sub + 4 * kg;        // 8 consecutive features from n
                    if (n < a.N) {
                        if constexpr (EPI == EPI_HEADS) {
                            const HeadsOut& ho = a.heads;
                            const int ri = n / ho.inner, nn = n - ri * ho.inner;
                            heads_store(ho, ri, nn >> 6, m, nn & 63, ch);
                        } else {
                            st16(a.out_pk + pk_off(m, n, a.N), ch);
                        }
                    }
                }
            }
          }
        }
        __syncthreads();
    }
}
// One-row-tile form of rows_block16 split by token group: a unit = (16 output features, 16 of the 32 rows); the two units
// of a feature slice are `pair_stride` apart in the unit numbering (callers place them on the same XCD).  Halves the
// activation bytes a workgroup pulls through its CU's L1 (see gemm_rows_resid_split_kernel).
template <int EPI, int NW, int U>
MG_DEV void rows_split_block(const GemmArgs& a, int ht, int g, char* smem) {
    static_assert(EPI == EPI_PK || EPI == EPI_PK_RELU || EPI == EPI_HEADS, "packed / per-head epilogues only");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int nt = ht >> 1, sub = ht & 1;
    const int kt16 = a.K >> 4, kp = kt16 >> 1;
    const int per = (kp + NW - 1) / NW;
    const int p0 = w * per, p1 = (p0 + per) < kp ? (p0 + per) : kp;
    float* rsl = (float*)(smem + NW * 4 * 64 * sizeof(float));     // [32]
    const size_t lane_off = (size_t)(kg >> 1) * TILE_BYTES + (size_t)(kg & 1) * 512;
    const char* wp = (const char*)(a.W + pk_tile_off(nt, 0, a.K)) + lane_off + (size_t)(16 * sub + r16) * 16;
    const char* xp = (const char*)a.X + (size_t)a.x_k0 * TILE_BYTES + lane_off + (size_t)(16 * g + r16) * 16;
    RsRegs rsr;
    rs_issue(a.rs, a.M, 32, tid, NW * 64, rsr);
    int p = p0;
    uint4 wf[U], xf[U];
    const bool full = p + U <= p1;
    if (full) {
#pragma unroll
        for (int u = 0; u < U; ++u) wf[u] = ld16_stream(wp + (size_t)(p + u) * (2 * TILE_BYTES));
#pragma unroll
        for (int u = 0; u < U; ++u) xf[u] = ld16(xp + (size_t)(2 * (p + u)) * TILE_BYTES);
    }
    rs_finish(a.rs, a.M, 32, rsl, tid, NW * 64, rsr);
    f32x4 acc = acc4_zero();
    if (full) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc = mfma16(wf[u], xf[u], acc);
        p += U;
    }
    for (; p < p1; ++p) acc = mfma16(ld16_stream(wp + (size_t)p * (2 * TILE_BYTES)), ld16(xp + (size_t)(2 * p) * TILE_BYTES), acc);
    float* slab = (float*)smem;                                  // [NW][4][64]
#pragma unroll
    for (int j = 0; j < 4; ++j) slab[(w * 4 + j) * 64 + lane] = acc[j];
    __syncthreads();
    if (w == 0) {
        const int m = 16 * g + r16;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = 0.f;
            for (int ww = 0; ww < NW; ++ww) t += slab[(ww * 4 + j) * 64 + lane];
            t *= rsl[m];
            v[j] = (EPI == EPI_PK_RELU) ? fmaxf(t, 0.f) : t;
        }
        const uint32_t lo = pack_bf16(v[0], v[1]), hi = pack_bf16(v[2], v[3]);
        const uint32_t plo = __shfl_xor(lo, 16), phi = __shfl_xor(hi, 16);
        if ((kg & 1) == 0 && m < a.M) {
            const uint4 ch = make_uint4(lo, hi, plo, phi);
            const int n = nt * 32 + 16 * sub + 4 * kg;
            if (n < a.N) {
                if constexpr (EPI == EPI_HEADS) {
                    const HeadsOut& ho = a.heads;
                    const int ri = n / ho.inner, nn = n - ri * ho.inner;
                    heads_store(ho, ri, nn >> 6, m, nn & 63, ch);
                } else {
                    st16(a.out_pk + pk_off(m, n, a.N), ch);
                }
            }
        }
    }
}
template <int EPI, int MT, bool HALF, int NW = 4, int FT = 1>
__global__ __launch_bounds__(NW * 64) void gemm_rows_kernel(GemmArgs a) {
    MG_DYN_SMEM(smem);
    if constexpr (HALF && (EPI == EPI_PK || EPI == EPI_PK_RELU || EPI == EPI_HEADS || EPI == EPI_PK_SWIGLU || EPI == EPI_F32_STORE)) rows_block16<EPI, MT, NW, 4, FT>(a, blockIdx.x, smem);
    else rows_block<EPI, MT, HALF, NW, 8>(a, blockIdx.x, smem);
}

// row-tile split forms (grid.y = row tile, one tile per workgroup): separate kernels, so that the argument structs of the
// ordinary forms stay read-only (modifying the by-value arguments in place cost the per-head QKV projection 5.2 -> 10.5 us)
template <int EPI, bool HALF, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_rows_split_kernel(GemmArgs a) {
    MG_DYN_SMEM(smem);
    shift_rows<EPI>(a, blockIdx.y);
    if constexpr (HALF && (EPI == EPI_PK || EPI == EPI_PK_RELU || EPI == EPI_HEADS || EPI == EPI_PK_SWIGLU || EPI == EPI_F32_STORE)) rows_block16<EPI, 1, NW, 4>(a, blockIdx.x, smem);
    else rows_block<EPI, 1, HALF, NW, 8>(a, blockIdx.x, smem);
}

// half-tile projections with >= 3 row tiles taking both halves per workgroup (same bits): -1 = as the call asks (GemmArgs::both_halves: the
// engine sets it on contexts that share the GPU - +2 % in flight, -2.5 % for a call alone), 0 never, 1 always (gemm_rows_set_ft2 / MG_ROWS_FT2)
static int g_rows_ft2 = [] { const char* e = getenv("MG_ROWS_FT2"); return (e && (e[0] == '0' || e[0] == '1')) ? e[0] - '0' : -1; }();
void gemm_rows_set_ft2(int mode) { g_rows_ft2 = mode < 0 ? -1 : (mode ? 1 : 0); }
static bool rows_ft2(const GemmArgs& a) { return g_rows_ft2 < 0 ? a.both_halves != 0 : g_rows_ft2 != 0; }
template <int EPI>
static void gemm_rows_mt(const GemmArgs& a, int mt, bool half, mgStream_t stream) {
    // half-tile projections (few workgroups, latency-bound): 8 waves split K so each wave's share is one load round
    const int NW = half ? 8 : 4;
    const int mts = rows_split_tiles(mt, (size_t)a.N * a.K);
    const bool split = mts < mt;
    const dim3 grid(((a.N + 31) / 32) * (half ? 2 : 1), split ? (mt + mts - 1) / mts : 1), block(NW * 64);
    if (split) mt = mts;
    const size_t sh = (size_t)NW * 16 * 64 * sizeof(float) + (size_t)32 * mt * sizeof(float);
    if (split) {
        if (half) MG_LAUNCH((gemm_rows_split_kernel<EPI, true, 8>), grid, block, sh, stream, a);
        else MG_LAUNCH((gemm_rows_split_kernel<EPI, false, 4>), grid, block, sh, stream, a);
        return;
    }
    // from three row tiles on a half-tile workgroup takes both halves of its weight tile (rows_block16 FT = 2: half the activation
    // re-reads from L2, half the workgroups; bit-identical)
    if constexpr (EPI == EPI_PK || EPI == EPI_PK_RELU || EPI == EPI_HEADS || EPI == EPI_F32_STORE) {
        if (half && mt >= 3 && rows_ft2(a) && (a.N % 32) == 0) {
            const dim3 grid2((a.N + 31) / 32);
#define MG_GR2(MTV) case MTV: MG_LAUNCH((gemm_rows_kernel<EPI, MTV, true, 8, 2>), grid2, block, sh, stream, a); return;
            switch (mt) { MG_GR2(3) MG_GR2(4) MG_GR2(5) MG_GR2(6) MG_GR2(7) MG_GR2(8) default: break; }
#undef MG_GR2
        }
    }
#define MG_GR(MTV)                                                                                   \
    case MTV:                                                                                        \
        if (half) MG_LAUNCH((gemm_rows_kernel<EPI, MTV, true, 8>), grid, block, sh, stream, a);      \
        else MG_LAUNCH((gemm_rows_kernel<EPI, MTV, false, 4>), grid, block, sh, stream, a);          \
        break;
    switch (mt) {
        MG_GR(1) MG_GR(2) MG_GR(3) MG_GR(4) MG_GR(5) MG_GR(6) MG_GR(7) MG_GR(8)
        default: break;
    }
#undef MG_GR
}

// half-tile form only (epilogues that exist there alone)
template <int EPI>
static void gemm_rows_mt_half(const GemmArgs& a, int mt, mgStream_t stream) {
    const int mts = rows_split_tiles(mt, (size_t)a.N * a.K);
    const bool split = mts < mt;
    const dim3 grid(((a.N + 31) / 32) * 2, split ? (mt + mts - 1) / mts : 1), block(8 * 64);
    if (split) mt = mts;
    const size_t sh = (size_t)8 * 16 * 64 * sizeof(float) + (size_t)32 * mt * sizeof(float);
    if (split) { MG_LAUNCH((gemm_rows_split_kernel<EPI, true, 8>), grid, block, sh, stream, a); return; }
#define MG_GR(MTV) case MTV: MG_LAUNCH((gemm_rows_kernel<EPI, MTV, true, 8>), grid, block, sh, stream, a); break;
    switch (mt) {
        MG_GR(1) MG_GR(2) MG_GR(3) MG_GR(4) MG_GR(5) MG_GR(6) MG_GR(7) MG_GR(8)
        default: break;
    }
#undef MG_GR
}

// ---------------------------------------------------------------------------------------------------------
// decode-step GEMM, split-K form: grid = (N/32) x KS workgroups so that EVERY CU streams a share of the weights
// (a 1024-wide projection has only 32 feature tiles).  Each workgroup reduces its 4 waves through LDS and stores
// its fp32 32x32 partial tile to slab P[ks] with plain coalesced stores; the consumer kernel (fused residual-add +
// RMSNorm, the single-query attention kernels, relu_pack) sums the KS slabs in a fixed order -> deterministic, no
// atomics, no extra reduction launch.
// ---------------------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void gemm_rows_splitk_kernel(const uint16_t* X, const uint16_t* W, float* P, int M, int N, int K, int ldp,
                                                          size_t slab_stride, int KS, RowScale rs, TopOut top) {
    MG_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float* rsl = (float*)(smem + 4 * 16 * 64 * sizeof(float));     // [32*MT] deferred RMSNorm scale per row
    RsRegs rsr;                                                    // load order = wait order (see resid_block16)
    rs_issue(rs, M, 32 * MT, tid, 256, rsr);
    const int ntiles = (N + 31) >> 5;
    const int nt = blockIdx.x % ntiles, ks = blockIdx.x / ntiles;
    const int kt16 = K >> 4;
    const int per_blk = (kt16 + KS - 1) / KS;
    const int kb0 = ks * per_blk, kb1 = (kb0 + per_blk) < kt16 ? (kb0 + per_blk) : kt16;
    const int per = (kb1 - kb0 + 3) >> 2;
    const int k0 = kb0 + w * per, k1 = (k0 + per) < kb1 ? (k0 + per) : kb1;

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = acc_zero();
    const char* wp = (const char*)(W + pk_tile_off(nt, 0, K)) + lane * 16;
    const char* xp = (const char*)X + lane * 16;
    constexpr int U = MT <= 2 ? 16 : 8;           // K = 1024 over 4 waves: 16 k-tiles per wave in one round
    int kt = k0;
    uint4 wf[U];
    const bool first_full = kt + U <= k1;
    if (first_full) {
#pragma unroll
        for (int u = 0; u < U; ++u) wf[u] = ld16_stream(wp + (size_t)(kt + u) * TILE_BYTES);
    }
    rs_finish(rs, M, 32 * MT, rsl, tid, 256, rsr);
    for (bool first = true; kt + U <= k1; kt += U, first = false) {
        if (!first) {
#pragma unroll
            for (int u = 0; u < U; ++u) wf[u] = ld16_stream(wp + (size_t)(kt + u) * TILE_BYTES);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
                acc[i] = mfma32(ld16(xp + ((size_t)i * kt16 + (kt + u)) * TILE_BYTES), wf[u], acc[i]);
        }
    }
    for (; kt < k1; ++kt) {
        const uint4 wf = ld16_stream(wp + (size_t)kt * TILE_BYTES);
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = mfma32(ld16(xp + ((size_t)i * kt16 + kt) * TILE_BYTES), wf, acc[i]);
    }
    float* slab = (float*)smem;
    float* tl = (float*)(smem + 4 * 16 * 64 * sizeof(float) + 32 * MT * sizeof(float));      // [4 waves][8 rows][33] (TopOut only)
    float* out = P + (size_t)ks * slab_stride;
    const int half = lane >> 5, n = nt * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[(w * 16 + r) * 64 + lane] = acc[i][r];
        __syncthreads();
        // wave w finishes registers 4w..4w+3 of the tile (rows acc_row(4w+j, half)), all 4 waves store in parallel
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = w * 4 + j;
            float v = slab[(0 * 16 + r) * 64 + lane];
            v += slab[(1 * 16 + r) * 64 + lane];
            v += slab[(2 * 16 + r) * 64 + lane];
            v += slab[(3 * 16 + r) * 64 + lane];
            const int m = 32 * i + acc_row(r, half);
            const float lg = v * rsl[m < M ? m : 0];
            if (m < M && n < N && (!top.ptop || top.write_logits)) out[(size_t)m * ldp + n] = lg;
            if (top.ptop) {
                // stop tokens are kept out of the ranking, their logits stored apart; the row's 32 values of this workgroup go to
                // LDS (one padded line per row: this wave finishes 8 rows = 4 registers x 2 half-waves)
                const bool is_stop = n == top.stop[0] || n == top.stop[1] || n == top.stop[2] || n == top.stop[3];
                if (is_stop && m < M) {
                    const int k = n == top.stop[0] ? 0 : (n == top.stop[1] ? 1 : (n == top.stop[2] ? 2 : 3));
                    top.stopv[(size_t)m * 4 + k] = lg;
                }
                tl[(w * 8 + j * 2 + half) * 33 + (lane & 31)] = (n < N && !is_stop) ? lg : -3.0e38f;
            }
        }
        if (top.ptop) {
            // top-2 per row: lane t < 8 of the wave scans row t's 32 values serially (ascending index: a tie keeps the lower one,
            // as torch.argmax) - 32 LDS reads instead of 5 rounds of cross-lane exchanges per row
            __syncthreads();            // (the 8 rows a wave scans were written by that wave; a block barrier keeps the emulator build simple)
            if (lane < 8) {
                const float* rowv = tl + (w * 8 + lane) * 33;
                float b1 = -3.0e38f, b2 = -3.0e38f;
                int i1 = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const float x = rowv[k];
                    if (x > b1) { b2 = b1; b1 = x; i1 = k; }
                    else b2 = fmaxf(b2, x);
                }
                const int jj = lane >> 1, hh = lane & 1;
                const int m = 32 * i + acc_row(w * 4 + jj, hh);
                if (m < M) top.ptop[(size_t)m * ntiles + nt] = make_float4(b1, b2, __int_as_float(b1 > -3.0e38f ? nt * 32 + i1 : 0x7fffffff), 0.f);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// residual projection of the decode step + next RMSNorm folded in (see mg_kernels.h gemm_rows_resid)
// workgroup = NW waves, 8 output features (a quarter weight tile; other lanes feed zeros), all M rows.
// ---------------------------------------------------------------------------------------------------------
template <int MT, int NW>
MG_DEV void resid_block(const ResidArgs& a, int bid, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int nt = bid >> 2, sub = bid & 3;
    const bool wvalid = (l32 >> 3) == sub;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    const int M = a.M, N = a.N;
    const int kt16 = a.K >> 4;
    const int per = (kt16 + NW - 1) / NW;
    const int k0 = w * per, k1 = (k0 + per) < kt16 ? (k0 + per) : kt16;
    float* rsl = (float*)(smem + NW * 4 * 64 * sizeof(float));     // [32*MT]
    const char* wp = (const char*)(a.W + pk_tile_off(nt, 0, a.K)) + lane * 16;
    const int xkts = a.x_kts ? a.x_kts : kt16;
    const char* xp = (const char*)a.X + ((size_t)a.x_k0 * TILE_BYTES + lane * 16);
    // the wide (K = d_ff) form keeps its whole K share in flight at once - while the accumulators of the live m-tiles
    // leave room for it (1024 threads: 128 registers per lane; with 3+ m-tiles sixteen fragments spill to scratch)
    constexpr int U = (NW >= 16 && MT <= 2) ? 16 : 8;
    // first round of the weight stream (HBM) first; the row scales' and the residual's L2 round trips overlap it
    int kt = k0;
    uint4 wf[U];
    const bool first_full = kt + U <= k1;
    if (first_full) {
#pragma unroll
        for (int u = 0; u < U; ++u) wf[u] = wvalid ? ld16_stream(wp + (size_t)(kt + u) * TILE_BYTES) : zero4;
    }
    // the wave that finishes m-tile i (i % NW == w; MT <= NW so at most one) fetches its slice of the residual now
    const int n0 = nt * 32 + sub * 8 + half * 4;
    const int my_i = w < MT ? w : -1;
    float4 h_pre = make_float4(0.f, 0.f, 0.f, 0.f);
    if (my_i >= 0 && 32 * my_i + l32 < M) h_pre = *(const float4*)(a.h + (size_t)(32 * my_i + l32) * N + n0);
    block_row_scales(a.rs, M, 32 * MT, rsl, tid, NW * 64);
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = acc_zero();
    for (bool first = true; kt + U <= k1; kt += U, first = false) {
        if (!first) {
#pragma unroll
            for (int u = 0; u < U; ++u) wf[u] = wvalid ? ld16_stream(wp + (size_t)(kt + u) * TILE_BYTES) : zero4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = mfma32(wf[u], ld16(xp + ((size_t)i * xkts + (kt + u)) * TILE_BYTES), acc[i]);
        }
    }
    for (; kt < k1; ++kt) {
        const uint4 wf = wvalid ? ld16_stream(wp + (size_t)kt * TILE_BYTES) : zero4;
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = mfma32(wf, ld16(xp + ((size_t)i * xkts + kt) * TILE_BYTES), acc[i]);
    }
    // D rows = features of the tile; the valid 8 (8*sub .. +7) sit in registers 4*sub .. 4*sub+3:
    // lane (row m = l32, half) holds features 8*sub + 4*half + j.  Reduce those 4 registers over the NW waves.
    float* slab = (float*)smem;                         // [NW][4][64]
    const int nparts = N >> 3;
    const int x_ld = a.x_ld ? a.x_ld : N, x2_ld = a.x2_ld ? a.x2_ld : N;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) if (r == 4 * sub + j) v = acc[i][r];     // sub is workgroup-uniform
            slab[(w * 4 + j) * 64 + lane] = v;
        }
        __syncthreads();
        if ((i % NW) == w) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = 0.f;
                for (int ww = 0; ww < NW; ++ww) t += slab[(ww * 4 + j) * 64 + lane];
                v[j] = t * rsl[32 * i + l32];
            }
            const int m = 32 * i + l32;
            float ss = 0.f;
            if (m < M) {
                float4 hv = h_pre;
                hv.x += v[0]; hv.y += v[1]; hv.z += v[2]; hv.w += v[3];
                *(float4*)(a.h + (size_t)m * N + n0) = hv;
                ss = (hv.x * hv.x + hv.y * hv.y) + (hv.z * hv.z + hv.w * hv.w);
                if (a.x_pk) {
                    const float4 g = *(const float4*)(a.gain + n0);
                    const float gs = a.gscale;
                    *(uint2*)(a.x_pk + pk_off(m, a.x_col0 + n0, x_ld)) =
                        make_uint2(pack_bf16(hv.x * g.x * gs, hv.y * g.y * gs), pack_bf16(hv.z * g.z * gs, hv.w * g.w * gs));
                }
                if (a.x2_pk)
                    *(uint2*)(a.x2_pk + pk_off(m, a.x2_col0 + n0, x2_ld)) = make_uint2(pack_bf16(hv.x, hv.y), pack_bf16(hv.z, hv.w));
            }
            ss += __shfl_xor(ss, 32);
            if (m < M && half == 0) a.part[(size_t)m * nparts + bid] = ss;
        }
        __syncthreads();
    }
}
// The same residual projection on v_mfma_f32_16x16x32_bf16: the workgroup's 8 output features fill half of the 16
// feature rows of the tile (a quarter of the 32 rows of the 32x32x16 form), one MFMA spans two k-tiles and costs about
// half the matrix-pipe time, and one wave-load of weights carries 512 useful bytes instead of 256.
//   A operand (weights): lane (r16 = l%16, kg = l/16) <- chunk (row 8*sub + r16, k-half kg&1) of k-tile 2p + kg/2
//   B operand (tokens 16g .. 16g+15 of an m-tile): same chunk addressing on the activation tile
//   D: lane holds features 4*kg + j (valid: kg < 2) of token 16g + r16
// F16: the workgroup owns 16 output features (every row of the MFMA's A operand carries weights) instead of 8 - half the
// workgroups, each pulling the whole activation block once: half the activation bytes through L2 for the projection.  The forms
// with several row tiles use it (their time is the L2 -> CU traffic of the activations: 128 workgroups x 1 MB at 128 rows of the
// FFN output projection); the one-tile forms keep 8 features (more workgroups streaming weights: latency).  Each output element is
// the same chain of MFMAs and the partial sums of squares keep their 8-feature groups: bit-identical either way.
template <int MT, int NW, int U, bool TRACE = false, bool F16 = false>
MG_DEV void resid_block16(const ResidArgs& a, int bid, char* smem, long long* trace = nullptr) {
    static_assert(MT <= NW, "at most two finishing units per wave");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // TRACE (tools/trace_resid.py only): shader-clock stamps of wave phases -> trace[(bid*NW + w)*8 + k]
    auto stamp = [&](int k) {
#ifndef MG_EMU
        if constexpr (TRACE) { if (lane == 0) trace[((size_t)bid * NW + w) * 8 + k] = (long long)__builtin_readcyclecounter(); }
#endif
        (void)k;
    };
    stamp(0);
    const int r16 = lane & 15, kg = lane >> 4;
    const int nt = F16 ? bid >> 1 : bid >> 2, sub = F16 ? bid & 1 : bid & 3;
    const bool wvalid = F16 || r16 < 8;
    const bool kgvalid = F16 || kg < 2;                       // lanes whose 4 accumulator registers are output features
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    const int M = a.M, N = a.N;
    const int kt16 = a.K >> 4, kp = kt16 >> 1;                 // pairs of k-tiles
    const int per = (kp + NW - 1) / NW;
    const int p0 = w * per, p1 = (p0 + per) < kp ? (p0 + per) : kp;
    float* rsl = (float*)(smem + NW * 8 * 64 * sizeof(float));     // [32*MT]
    const size_t lane_off = (size_t)(kg >> 1) * TILE_BYTES + (size_t)(kg & 1) * 512;
    const char* wp = (const char*)(a.W + pk_tile_off(nt, 0, a.K)) + lane_off + (size_t)((F16 ? 16 : 8) * sub + r16) * 16;
    const int xkts = a.x_kts ? a.x_kts : kt16;
    const char* xp = (const char*)a.X + (size_t)a.x_k0 * TILE_BYTES + lane_off + (size_t)r16 * 16;
    // load order = wait order (the vector-memory queue retires in order): row-scale partials and the residual slice
    // first (small, L2), then the weight stream (HBM); the activation fragments follow in the MFMA loop
    RsRegs rsr;
    rs_issue(a.rs, M, 32 * MT, tid, NW * 64, rsr);
    const int n0 = nt * 32 + sub * (F16 ? 16 : 8) + kg * 4;     // this lane's 4 features (kgvalid)
    // the (m-tile, token group) unit this wave finishes: unit f = 2*i + g goes to wave f (2*MT <= NW), so the two token
    // groups of a tile are reduced and stored by two waves in parallel
    // (with more than NW/2 row tiles a wave takes a second unit, f + NW)
    float4 h_pre[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int f = w + q * NW;
        if (f < 2 * MT && kgvalid) {
            const int m = 32 * (f >> 1) + 16 * (f & 1) + r16;
            if (m < M) h_pre[q] = *(const float4*)(a.h + (size_t)m * N + n0);
        }
    }
    int p = p0;
    uint4 wf[U];
    // one row tile, 8 waves: the activation fragments of the first round are fetched up front as well (the 16-wave form
    // has 64 registers per lane, where that spills: measured 7.8 -> 15.5 us)
    constexpr bool XPF = MT == 1 && NW <= 8;
    uint4 xf[XPF ? U : 1][2];
    if (p + U <= p1) {
#pragma unroll
        for (int u = 0; u < U; ++u) wf[u] = wvalid ? ld16_stream(wp + (size_t)(p + u) * (2 * TILE_BYTES)) : zero4;
        if constexpr (XPF) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const char* xt = xp + (size_t)(2 * (p + u)) * TILE_BYTES;
                xf[u][0] = ld16(xt); xf[u][1] = ld16(xt + 256);
            }
        }
    }
    stamp(1);
    rs_finish(a.rs, M, 32 * MT, rsl, tid, NW * 64, rsr);
    stamp(2);
    f32x4 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc[i][0] = acc4_zero(); acc[i][1] = acc4_zero(); }
    for (bool first = true; p + U <= p1; p += U, first = false) {
        if (!first) {
#pragma unroll
            for (int u = 0; u < U; ++u) wf[u] = wvalid ? ld16_stream(wp + (size_t)(p + u) * (2 * TILE_BYTES)) : zero4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const char* xt = xp + ((size_t)i * xkts + 2 * (p + u)) * TILE_BYTES;
                if (XPF && first) {
                    acc[i][0] = mfma16(wf[u], xf[XPF ? u : 0][0], acc[i][0]);
                    acc[i][1] = mfma16(wf[u], xf[XPF ? u : 0][1], acc[i][1]);
                } else {
                    acc[i][0] = mfma16(wf[u], ld16(xt), acc[i][0]);
                    acc[i][1] = mfma16(wf[u], ld16(xt + 256), acc[i][1]);
                }
            }
        }
    }
    for (; p < p1; ++p) {
        const uint4 w1 = wvalid ? ld16_stream(wp + (size_t)p * (2 * TILE_BYTES)) : zero4;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const char* xt = xp + ((size_t)i * xkts + 2 * p) * TILE_BYTES;
            acc[i][0] = mfma16(w1, ld16(xt), acc[i][0]);
            acc[i][1] = mfma16(w1, ld16(xt + 256), acc[i][1]);
        }
    }
    if constexpr (TRACE) { if (acc[0][0][0] == 123456.f) a.part[0] = 0.f; }      // MFMA results are in
    stamp(3);
    // reduce the NW K-slices: slab[w][g*4 + j][lane]
    float* slab = (float*)smem;
    const int nparts = N >> 3;
    const int x_ld = a.x_ld ? a.x_ld : N, x2_ld = a.x2_ld ? a.x2_ld : N;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) slab[(w * 8 + g * 4 + j) * 64 + lane] = acc[i][g][j];
        __syncthreads();
        stamp(4);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = w + q * NW;
            if (f < 2 * MT && (f >> 1) == i) {
                const int g = f & 1;
                const int m = 32 * i + 16 * g + r16;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = 0.f;
                    for (int ww = 0; ww < NW; ++ww) t += slab[(ww * 8 + g * 4 + j) * 64 + lane];
                    v[j] = t * rsl[32 * i + 16 * g + r16];
                }
                float ss = 0.f;
                if (m < M && kgvalid) {
                    float4 hv = h_pre[q];
                    hv.x += v[0]; hv.y += v[1]; hv.z += v[2]; hv.w += v[3];
                    *(float4*)(a.h + (size_t)m * N + n0) = hv;
                    ss = (hv.x * hv.x + hv.y * hv.y) + (hv.z * hv.z + hv.w * hv.w);
                    if (a.x_pk) {
                        const float4 gn = *(const float4*)(a.gain + n0);
                        const float gs = a.gscale;
                        *(uint2*)(a.x_pk + pk_off(m, a.x_col0 + n0, x_ld)) =
                            make_uint2(pack_bf16(hv.x * gn.x * gs, hv.y * gn.y * gs), pack_bf16(hv.z * gn.z * gs, hv.w * gn.w * gs));
                    }
                    if (a.x2_pk)
                        *(uint2*)(a.x2_pk + pk_off(m, a.x2_col0 + n0, x2_ld)) = make_uint2(pack_bf16(hv.x, hv.y), pack_bf16(hv.z, hv.w));
                }
                ss += __shfl_xor(ss, 16);                        // features 0-3 (kg 0) + 4-7 (kg 1)  [F16: and 8-11 (kg 2) + 12-15 (kg 3)]
                if (F16) { if (m < M && (kg & 1) == 0) a.part[(size_t)m * nparts + 2 * bid + (kg >> 1)] = ss; }
                else if (m < M && kg == 0) a.part[(size_t)m * nparts + bid] = ss;
            }
            stamp(5);
        }
        __syncthreads();
    }
    stamp(6);
}
template <int MT, int NW, bool F16 = false>
__global__ __launch_bounds__(NW * 64) void gemm_rows_resid_kernel(ResidArgs a) {
    MG_DYN_SMEM(smem);
    // pairs of k-tiles per wave and round: 8 covers K = 4096 over 16 waves, 4 covers K = 1024 over 8 waves, in one round
    resid_block16<MT, NW, (NW >= 16 ? 8 : 4), false, F16>(a, blockIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------------------
// Residual projection over SEVERAL row tiles (64 .. 256 live rows), K-slab form.  The one-workgroup forms above give a workgroup
// a slice of output features and ALL of K, so every one of the N/16 workgroups pulls the whole activation block [rows][K] through
// its CU (1.3 MB at 160 rows of the FFN output projection; 84 MB of L2 -> CU traffic for 8.4 MB of weights) - and only N/16 CUs
// work.  Here the K CHUNKS that the waves of those forms take (chunk s = pairs of k-tiles [s per, (s+1) per)) become workgroups:
// workgroup (feature tile nt of 32, chunk s) reads the activation slab [rows][chunk] once for 32 features (two 16-feature MFMA
// tiles share every activation fragment), keeps its whole weight slab and its whole activation slab in flight at once (ONE memory
// round trip, no LDS, no barrier in the main phase), and leaves its partial sums p_s in kpart.  The LAST workgroup of a feature tile
// to arrive (ticket) adds p_0 .. p_{S-1} IN THAT ORDER - the order in which the one-workgroup forms add their waves' partials -
// applies the row scale and runs their epilogue: every output element is the same chain of MFMAs and the same chain of additions,
// i.e. the SAME BITS as gemm_rows_resid_kernel / gemm_rows_resid_split_kernel (tests/test_kernels.py), whatever the row count.
// A wave owns one 32-row tile; workgroups of a feature tile sit on one XCD (ids 8 apart) so the partials meet in one L2.
// ---------------------------------------------------------------------------------------------------------
template <int MT, int PER>
__global__ __launch_bounds__(64 * MT) void gemm_rows_resid_mt_kernel(ResidArgs a, int S, int fmode) {
    MG_DYN_SMEM(smem);
    float* rsl = (float*)smem;                                   // [32 * MT] (finishing workgroup)
    int* flag = (int*)(rsl + 32 * MT);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int M = a.M, N = a.N;
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int nt = (j / S) * 8 + xcd, s = j - (j / S) * S;       // feature tile (32 features), K chunk
    const int kt16 = a.K >> 4, xkts = a.x_kts ? a.x_kts : kt16;
    const int p0 = s * PER;
    const size_t lane_off = (size_t)(kg >> 1) * TILE_BYTES + (size_t)(kg & 1) * 512;
    const char* wp = (const char*)(a.W + pk_tile_off(nt, 0, a.K)) + lane_off + (size_t)r16 * 16 + (size_t)(2 * p0) * TILE_BYTES;
    const char* xp = (const char*)a.X + ((size_t)a.x_k0 + (size_t)w * xkts + 2 * p0) * TILE_BYTES + lane_off + (size_t)r16 * 16;
    uint4 wf[2][PER], xf[PER][2];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        wf[0][u] = ld16_stream(wp + (size_t)u * (2 * TILE_BYTES));
        wf[1][u] = ld16_stream(wp + (size_t)u * (2 * TILE_BYTES) + 256);          // features 16 .. 31 of the tile: rows 16 .. 31 of each k-half
        xf[u][0] = ld16(xp + (size_t)u * (2 * TILE_BYTES));
        xf[u][1] = ld16(xp + (size_t)u * (2 * TILE_BYTES) + 256);
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[ft][g] = acc4_zero();
#pragma unroll
    for (int u = 0; u < PER; ++u)
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int g = 0; g < 2; ++g) acc[ft][g] = mfma16(wf[ft][u], xf[u][g], acc[ft][g]);
    // partial sums: lane (token r16 of group g, kg) holds features 16 ft + 4 kg + j of row 32 w + 16 g + r16
    const int Mp = 32 * MT;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float* dst = a.kpart + ((size_t)s * Mp + (size_t)(32 * w + 16 * g + r16)) * N + nt * 32 + 16 * ft + 4 * kg;
            *(float4*)dst = make_float4(acc[ft][g][0], acc[ft][g][1], acc[ft][g][2], acc[ft][g][3]);
        }
    // hand-off to the last arrival (cdna_hip_programming.md, in-launch split-K reduction): every wave drains its stores, ONE lane
    // releases at agent scope (the XCD's L2 writes its dirty lines back: the L2s of the 8 XCDs are not coherent with each other) and
    // takes the ticket; the workgroup that draws S - 1 acquires once (its L1 may hold the partials of the previous launch) and reads
    if (fmode == 6) return;                                      // (timing experiments, MG_MT_FENCE: 6 = main phase only, 5 = no finishing)
#ifndef MG_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    if (tid == 0) {
#ifndef MG_EMU
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *flag = __hip_atomic_fetch_add(a.ticket + nt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        *flag = atomicAdd(a.ticket + nt, 1);
#endif
    }
    __syncthreads();
    if (*flag != S - 1) return;
    if (fmode == 5) { if (tid == 0) a.ticket[nt] = 0; return; }
    if (tid == 0) {
#ifndef MG_EMU
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        a.ticket[nt] = 0;                                        // (nobody else touches it before the next launch)
    }
    __syncthreads();
    // ---- last arrival: sum the chunks in order, then the epilogue of resid_block16 (F16 form: 16-feature blocks 2 nt, 2 nt + 1) ----
    block_row_scales(a.rs, M, Mp, rsl, tid, 64 * MT);
    __syncthreads();
    const int nparts = N >> 3;
    const int x_ld = a.x_ld ? a.x_ld : N, x2_ld = a.x2_ld ? a.x2_ld : N;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int m = 32 * w + 16 * g + r16;
            const int n0 = nt * 32 + 16 * ft + 4 * kg;
            const float* src = a.kpart + (size_t)m * N + n0;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            for (int s0 = 0; s0 < S; s0 += 8) {                  // the loads of 8 chunks in flight together, added in chunk order
                float4 pv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) pv[q] = *(const float4*)(src + (size_t)(s0 + q < S ? s0 + q : S - 1) * Mp * N);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (s0 + q < S) { v[0] += pv[q].x; v[1] += pv[q].y; v[2] += pv[q].z; v[3] += pv[q].w; }
            }
            const float rs = rsl[m];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] *= rs;
            float ss2 = 0.f;
            if (m < M) {
                float4 hv = *(const float4*)(a.h + (size_t)m * N + n0);
                hv.x += v[0]; hv.y += v[1]; hv.z += v[2]; hv.w += v[3];
                *(float4*)(a.h + (size_t)m * N + n0) = hv;
                ss2 = (hv.x * hv.x + hv.y * hv.y) + (hv.z * hv.z + hv.w * hv.w);
                if (a.x_pk) {
                    const float4 gn = *(const float4*)(a.gain + n0);
                    const float gs = a.gscale;
                    *(uint2*)(a.x_pk + pk_off(m, a.x_col0 + n0, x_ld)) =
                        make_uint2(pack_bf16(hv.x * gn.x * gs, hv.y * gn.y * gs), pack_bf16(hv.z * gn.z * gs, hv.w * gn.w * gs));
                }
                if (a.x2_pk)
                    *(uint2*)(a.x2_pk + pk_off(m, a.x2_col0 + n0, x2_ld)) = make_uint2(pack_bf16(hv.x, hv.y), pack_bf16(hv.z, hv.w));
            }
            ss2 += __shfl_xor(ss2, 16);                          // features 0-3 (kg 0) + 4-7 (kg 1), 8-11 (kg 2) + 12-15 (kg 3)
            if (m < M && (kg & 1) == 0) a.part[(size_t)m * nparts + 2 * (2 * nt + ft) + (kg >> 1)] = ss2;
        }
}

// Second launch of the K-slab form's TWO-LAUNCH variant (mode 2): the partial sums gemm_rows_resid_mt_kernel left in kpart (it returns after its
// stores: no ticket, no fences - the launch boundary publishes them) are added in chunk order and the epilogue of resid_block16 runs, spread
// over the chip: workgroup = (feature tile of 32, 32-row tile), wave = (16-feature half, token group).  Same additions in the same order as the
// last-arrival code above and as the one-workgroup forms: same bits.
__global__ __launch_bounds__(256) void gemm_rows_resid_merge_kernel(ResidArgs a, int S, int Mp) {
    MG_DYN_SMEM(smem);
    float* rsl = (float*)smem;                                   // [32]: row scales of this row tile
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int ft = wv >> 1, g = wv & 1;
    const int nt = blockIdx.x, w = blockIdx.y;
    const int M = a.M, N = a.N;
    const int m = 32 * w + 16 * g + r16;
    const int n0 = nt * 32 + 16 * ft + 4 * kg;
    const float* src = a.kpart + (size_t)m * N + n0;
    float4 pv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) pv[q] = *(const float4*)(src + (size_t)(q < S ? q : S - 1) * Mp * N);
    float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < M) hv = *(const float4*)(a.h + (size_t)m * N + n0);
    {   // row scales of the tile's 32 rows (the rows of the other tiles are not needed here)
        RowScale rs = a.rs;
        if (rs.part) rs.part += (size_t)32 * w * rs.nparts;
        block_row_scales(rs, M - 32 * w, 32, rsl, tid, 256);
    }
    __syncthreads();
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 16; ++q)
        if (q < S) { v[0] += pv[q].x; v[1] += pv[q].y; v[2] += pv[q].z; v[3] += pv[q].w; }
    const float rs = rsl[16 * g + r16];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] *= rs;
    const int nparts = N >> 3;
    const int x_ld = a.x_ld ? a.x_ld : N, x2_ld = a.x2_ld ? a.x2_ld : N;
    float ss2 = 0.f;
    if (m < M) {
        hv.x += v[0]; hv.y += v[1]; hv.z += v[2]; hv.w += v[3];
        *(float4*)(a.h + (size_t)m * N + n0) = hv;
        ss2 = (hv.x * hv.x + hv.y * hv.y) + (hv.z * hv.z + hv.w * hv.w);
        if (a.x_pk) {
            const float4 gn = *(const float4*)(a.gain + n0);
            const float gs = a.gscale;
            *(uint2*)(a.x_pk + pk_off(m, a.x_col0 + n0, x_ld)) =
                make_uint2(pack_bf16(hv.x * gn.x * gs, hv.y * gn.y * gs), pack_bf16(hv.z * gn.z * gs, hv.w * gn.w * gs));
        }
        if (a.x2_pk)
            *(uint2*)(a.x2_pk + pk_off(m, a.x2_col0 + n0, x2_ld)) = make_uint2(pack_bf16(hv.x, hv.y), pack_bf16(hv.z, hv.w));
    }
    ss2 += __shfl_xor(ss2, 16);
    if (m < M && (kg & 1) == 0) a.part[(size_t)m * nparts + 2 * (2 * nt + ft) + (kg >> 1)] = ss2;
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void gemm_rows_resid_rowsplit_kernel(ResidArgs a) {     // grid.y = row tile (shift_rows)
    MG_DYN_SMEM(smem);
    shift_rows(a, blockIdx.y);
    resid_block16<1, NW, (NW >= 16 ? 8 : 4)>(a, blockIdx.x, smem);
}

// FFN-wo form (K = d_ff) for ONE row tile, split by token group: a workgroup owns 8 output features and 16 of the 32
// rows, so that 2*N/8 workgroups cover all 256 CUs and each pulls 64 KB of weights + 128 KB of activations through its
// CU's L1 instead of 64 + 256 KB (the per-CU ingest at 64 B/clk is what bounds this projection).  The
// two workgroups of a feature slice sit 8 apart in the grid = on the same XCD, so the slice's weights reach that L2 once.
template <int NW, int U>
__global__ __launch_bounds__(NW * 64) void gemm_rows_resid_split_kernel(ResidArgs a) {
    MG_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int bid = blockIdx.x;
    const int g = (bid >> 3) & 1, slice = (bid >> 4) * 8 + (bid & 7);
    const int nt = slice >> 2, sub = slice & 3;
    const bool wvalid = r16 < 8;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    const int M = a.M, N = a.N;
    const int kt16 = a.K >> 4, kp = kt16 >> 1;
    const int per = (kp + NW - 1) / NW;
    const int p0 = w * per, p1 = (p0 + per) < kp ? (p0 + per) : kp;
    float* rsl = (float*)(smem + NW * 4 * 64 * sizeof(float));     // [32]
    const size_t lane_off = (size_t)(kg >> 1) * TILE_BYTES + (size_t)(kg & 1) * 512;
    const char* wp = (const char*)(a.W + pk_tile_off(nt, 0, a.K)) + lane_off + (size_t)(8 * sub + r16) * 16;
    const int xkts = a.x_kts ? a.x_kts : kt16;
    (void)xkts;
    const char* xp = (const char*)a.X + (size_t)a.x_k0 * TILE_BYTES + lane_off + (size_t)(16 * g + r16) * 16;
    RsRegs rsr;
    rs_issue(a.rs, M, 32, tid, NW * 64, rsr);
    const int n0 = nt * 32 + sub * 8 + kg * 4;
    const int m = 16 * g + r16;
    float4 h_pre = make_float4(0.f, 0.f, 0.f, 0.f);
    if (w == 0 && kg < 2 && m < M) h_pre = *(const float4*)(a.h + (size_t)m * N + n0);
    int p = p0;
    uint4 wf[U];
    if (p + U <= p1) {
#pragma unroll
        for (int u = 0; u < U; ++u) wf[u] = wvalid ? ld16_stream(wp + (size_t)(p + u) * (2 * TILE_BYTES)) : zero4;
    }
    rs_finish(a.rs, M, 32, rsl, tid, NW * 64, rsr);
    f32x4 acc = acc4_zero();
    for (bool first = true; p + U <= p1; p += U, first = false) {
        if (!first) {
#pragma unroll
            for (int u = 0; u < U; ++u) wf[u] = wvalid ? ld16_stream(wp + (size_t)(p + u) * (2 * TILE_BYTES)) : zero4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = mfma16(wf[u], ld16(xp + (size_t)(2 * (p + u)) * TILE_BYTES), acc);
    }
    for (; p < p1; ++p) {
        const uint4 w1 = wvalid ? ld16_stream(wp + (size_t)p * (2 * TILE_BYTES)) : zero4;
        acc = mfma16(w1, ld16(xp + (size_t)(2 * p) * TILE_BYTES), acc);
    }
    float* slab = (float*)smem;                                  // [NW][4][64]
#pragma unroll
    for (int j = 0; j < 4; ++j) slab[(w * 4 + j) * 64 + lane] = acc[j];
    __syncthreads();
    if (w == 0) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = 0.f;
            for (int ww = 0; ww < NW; ++ww) t += slab[(ww * 4 + j) * 64 + lane];
            v[j] = t * rsl[m];
        }
        const int nparts = N >> 3;
        const int x_ld = a.x_ld ? a.x_ld : N, x2_ld = a.x2_ld ? a.x2_ld : N;
        float ss = 0.f;
        if (m < M && kg < 2) {
            float4 hv = h_pre;
            hv.x += v[0]; hv.y += v[1]; hv.z += v[2]; hv.w += v[3];
            *(float4*)(a.h + (size_t)m * N + n0) = hv;
            ss = (hv.x * hv.x + hv.y * hv.y) + (hv.z * hv.z + hv.w * hv.w);
            if (a.x_pk) {
                const float4 gn = *(const float4*)(a.gain + n0);
                const float gs = a.gscale;
                *(uint2*)(a.x_pk + pk_off(m, a.x_col0 + n0, x_ld)) =
                    make_uint2(pack_bf16(hv.x * gn.x * gs, hv.y * gn.y * gs), pack_bf16(hv.z * gn.z * gs, hv.w * gn.w * gs));
            }
            if (a.x2_pk)
                *(uint2*)(a.x2_pk + pk_off(m, a.x2_col0 + n0, x2_ld)) = make_uint2(pack_bf16(hv.x, hv.y), pack_bf16(hv.z, hv.w));
        }
        ss += __shfl_xor(ss, 16);
        if (m < M && kg == 0) a.part[(size_t)m * nparts + slice] = ss;
    }
}

#ifdef MG_TOOLS
// instrumented copy of the 16-wave, one-row-tile form (tools/trace_resid.py)
__global__ __launch_bounds__(1024) void gemm_rows_resid_trace_kernel(ResidArgs a, long long* trace) {
    MG_DYN_SMEM(smem);
    resid_block16<1, 16, 8, true>(a, blockIdx.x, smem, trace);
}
void gemm_rows_resid_trace(const ResidArgs& r, long long* trace, mgStream_t stream) {
    const size_t sh = (size_t)16 * 8 * 64 * sizeof(float) + 32 * sizeof(float);
    MG_LAUNCH(gemm_rows_resid_trace_kernel, dim3(r.N / 8), dim3(1024), sh, stream, r, trace);
}
#endif

static std::atomic<int> g_resid_f16{1};
// K-slab form of the projections with several row tiles: OFF by default.  Measured (profiles/r05_e_rows_mt_kslab.txt, FFN output projection at 160
// rows, launches back to back): one-workgroup form 15.8 us; K-slab form 6.8 us for the main phase (activation traffic through L2 halved, all
// CUs at work) + 12 us for 512 workgroups' agent-scope releases (each XCD L2 writes back the partial sums just stored) + 23 us for the last
// arrival of a feature tile reading 16 chunks x 160 rows x 32 features of partials (327 KB through one CU) = 42 us; in flight 147.5 -> 126
// images/s.  The chunk count is fixed by the summation order every row count shares (16 wave partials added in order), so the finisher's
// read volume cannot shrink without changing every form's bits.  Kept (tests keep it bit-identical; MG_ROWS_MT=1 / mgk_set_rows_mt) as the
// starting point; the form with a second, chip-wide merge launch instead of tickets and fences (g_rows_mt = 2, gemm_rows_resid_merge_kernel) runs the
// same projection in 12.1 us and a call alone 4 % faster (decode step 4.87 -> 4.68 ms, beam queue on one context 25.3 -> 26.5 images/s), but with four
// contexts in flight the headline loses 1.3 % (147.5 -> 145.5): it writes and re-reads 10.5 MB of partial sums per launch, and in flight the bytes a
// launch moves are what it costs.  Off by default as well.
static std::atomic<int> g_rows_mt{0};
void gemm_rows_set_mt(int on) { g_rows_mt = on; }
void gemm_rows_set_resid_f16(int on) { g_resid_f16 = on; }
void gemm_rows_resid(const ResidArgs& r, mgStream_t stream) {
    int mt = (r.M + 31) / 32;
    const int mts = rows_split_tiles(mt, (size_t)r.N * r.K);                  // row tiles per workgroup, grid.y = groups (shift_rows)
    const bool split = mts < mt;
    const dim3 grid(r.N / 8, split ? (mt + mts - 1) / mts : 1);
    if (split) mt = mts;
    // 16 waves for the long K = d_ff stream up to 4 row tiles (the waves partition K: the same count for every row count a greedy
    // call can have, so that a row's sums do not depend on how many batches share its call; 8 accumulator registers per row tile in
    // the 16x16x32 form); the decode steps ask for 8 = every call size (ResidArgs::wide_tiles; beam-5 at batch 32 = 160 rows: 29.9 -> 22.8 us alone, 111 -> 115
    // images/s in flight; the 32x32x16 form of round 1 spilled there: 80 accumulator registers, 52 us).
    const bool wide = r.K > 2048 && mt <= (r.wide_tiles > 4 ? r.wide_tiles : 4);
    if (wide && mt == 1 && !split && r.M > 16 && ((r.N >> 3) & 7) == 0 && r.x_kts == 0) {      // one row tile: split by token group
        const size_t shs = (size_t)16 * 4 * 64 * sizeof(float) + 32 * sizeof(float);
        MG_LAUNCH((gemm_rows_resid_split_kernel<16, 8>), dim3(2 * (r.N >> 3)), dim3(1024), shs, stream, r);
        return;
    }
    const int NW = wide ? 16 : 8;
    const dim3 block(NW * 64);
    const size_t sh = (size_t)NW * 8 * 64 * sizeof(float) + (size_t)32 * mt * sizeof(float);
    if (split) {
        if (wide) MG_LAUNCH((gemm_rows_resid_rowsplit_kernel<16>), grid, block, sh, stream, r);
        else MG_LAUNCH((gemm_rows_resid_rowsplit_kernel<8>), grid, block, sh, stream, r);
        return;
    }
    // several row tiles, K-slab form: the chunks of the one-workgroup form's waves as workgroups (same bits; see gemm_rows_resid_mt_kernel)
    { static bool env_read = false; if (!env_read) { env_read = true; if (const char* e = getenv("MG_ROWS_MT")) g_rows_mt = atoi(e); } }   // A/B runs
    // MG_ROWS_MT_ALONE=1: a call that has the GPU to itself (ResidArgs::alone: no other execution context in flight) takes the two-launch K-slab form
    // from 4 row tiles on.  The launch is 23 % (160 rows) to 35 % (256 rows) faster back to back, but inside a decode step the gain is 0.5 - 3.5 % of
    // the step depending on the box and the cross-attention launch of the same call was timed 7 % slower beside it (103 -> 111 us, profiles/r05_e_*):
    // off by default.
    static int alone_env = -1;
    if (alone_env < 0) { const char* e = getenv("MG_ROWS_MT_ALONE"); alone_env = e ? atoi(e) : 0; }
    const int rows_mt = g_rows_mt ? g_rows_mt.load() : ((r.alone && alone_env && mt >= 4) ? 2 : 0);
    if (r.kpart && r.ticket && mt >= 2 && !split && (r.N & 255) == 0 && rows_mt) {
        const int NWf = wide ? 16 : 8, kp = r.K >> 5, per = (kp + NWf - 1) / NWf;
        if (per * NWf == kp && (per == 8 || per == 4 || per == 2)) {
            const int S = NWf;
            static int fmode = -1;
            if (fmode < 0) { const char* e = getenv("MG_MT_FENCE"); fmode = e ? atoi(e) : 0; }
            const dim3 gridk((r.N / 32) * S), blockk(64 * mt);
            const size_t shk = (size_t)32 * mt * sizeof(float) + 16;
            const bool two = rows_mt == 2 && S <= 16;              // two-launch variant: partial sums, then a chip-wide merge launch
            const int fm = two ? 6 : fmode;
#define MG_RMT(MTV)                                                                                   \
    case MTV:                                                                                         \
        if (per == 8) MG_LAUNCH((gemm_rows_resid_mt_kernel<MTV, 8>), gridk, blockk, shk, stream, r, S, fm);       \
        else if (per == 4) MG_LAUNCH((gemm_rows_resid_mt_kernel<MTV, 4>), gridk, blockk, shk, stream, r, S, fm);  \
        else MG_LAUNCH((gemm_rows_resid_mt_kernel<MTV, 2>), gridk, blockk, shk, stream, r, S, fm);                \
        break;
            switch (mt) { MG_RMT(2) MG_RMT(3) MG_RMT(4) MG_RMT(5) MG_RMT(6) MG_RMT(7) MG_RMT(8) default: break; }
#undef MG_RMT
            if (two) MG_LAUNCH(gemm_rows_resid_merge_kernel, dim3(r.N / 32, mt), dim3(256), 32 * sizeof(float), stream, r, S, 32 * mt);
            return;
        }
    }
    // several row tiles: 16 features per workgroup (see resid_block16 F16); g_resid_f16 = 0: the 8-feature form (tests, A/B runs)
    const bool f16 = mt >= 2 && (r.N & 15) == 0 && g_resid_f16;
    const dim3 grid16(r.N / 16);
    // Measured and rejected (profiles/r04_r_resid_row_groups_rejected.txt): the row tiles of the long K = d_ff projection in groups over
    // grid.y (a workgroup pulls its rows and the weight slice through ONE CU's 64 B/clk: 1.1 MB at 128 rows) - alone the launch drops
    // from 20.5 to 12.9 (groups of 2 tiles) / 8.8 us (1 tile), with four contexts in flight the run is SLOWER (142.1 -> 141.2 / 137.6
    // images/s): in flight the bytes moved through L2 count, not a launch's latency, and the groups re-read the weight slice.
#define MG_RR(MTV)                                                                                 \
    case MTV:                                                                                      \
        if (f16 && wide) MG_LAUNCH((gemm_rows_resid_kernel<MTV, 16, true>), grid16, block, sh, stream, r);   \
        else if (f16) MG_LAUNCH((gemm_rows_resid_kernel<MTV, 8, true>), grid16, block, sh, stream, r);       \
        else if (wide) MG_LAUNCH((gemm_rows_resid_kernel<MTV, 16>), grid, block, sh, stream, r);   \
        else MG_LAUNCH((gemm_rows_resid_kernel<MTV, 8>), grid, block, sh, stream, r);              \
        break;
    switch (mt) {
        MG_RR(1) MG_RR(2) MG_RR(3) MG_RR(4) MG_RR(5) MG_RR(6) MG_RR(7) MG_RR(8)
        default: break;
    }
#undef MG_RR
}
void gemm_rows_resid(const uint16_t* X, const uint16_t* W, float* h, const float* gain, float gscale, uint16_t* x_pk, float* part,
                     int M, int N, int K, const RowScale& rs, mgStream_t stream) {
    ResidArgs r{};
    r.X = X; r.W = W; r.h = h; r.gain = gain; r.gscale = gscale; r.x_pk = x_pk; r.part = part; r.M = M; r.N = N; r.K = K; r.rs = rs;
    gemm_rows_resid(r, stream);
}

// Residual projection and a half-tile projection side by side in one grid (8 waves per workgroup; the second
// projection keeps up to 16 k-tiles per wave in flight: K = d_model + inner of the product weights in one round).
// [residual projection | half-tile projection split by token group], one row tile: nres + 2*(N2/16) workgroups laid out so
// that every aligned group of 16 consecutive workgroup ids holds 8 residual workgroups... (simple form: the second
// projection's units follow the residual ones; unit u -> slice (u>>4)*8 + (u&7), group (u>>3)&1: same XCD for a slice)
template <int EPI>
__global__ __launch_bounds__(512) void gemm_rows_pair_split_kernel(ResidArgs r, GemmArgs g, int nres) {
    MG_DYN_SMEM(smem);
    if ((int)blockIdx.x < nres) { resid_block16<1, 8, 4>(r, blockIdx.x, smem); return; }
    const int u = (int)blockIdx.x - nres;
    rows_split_block<EPI, 8, 8>(g, (u >> 4) * 8 + (u & 7), (u >> 3) & 1, smem);
}
template <int EPI, int MT, bool HALF, bool F16 = false, int FT = 1>
__global__ __launch_bounds__(512) void gemm_rows_pair_kernel(ResidArgs r, GemmArgs g, int nres) {
    MG_DYN_SMEM(smem);
    if ((int)blockIdx.x < nres) resid_block16<MT, 8, 4, false, F16>(r, blockIdx.x, smem);
    else if constexpr (HALF) rows_block16<EPI, MT, 8, 8, FT>(g, (int)blockIdx.x - nres, smem);     // K = d + inner: 8 pairs per wave
    else rows_block<EPI, MT, false, 8, 16>(g, (int)blockIdx.x - nres, smem);
}
void gemm_rows_pair(const ResidArgs& r, const GemmArgs& g, int epi, mgStream_t stream) {
    const int mt = (r.M + 31) / 32;       // (no row-tile split form: the pair projections exist for the main decoder's large weights only)
    // every workgroup of the second projection reads ALL of its activation window (rows x K) from L2: with many output
    // features (FFN wi: 128 tiles) whole 32-feature tiles halve that traffic (+0.8 % end to end), with few (cross-Q:
    // 32 tiles) half tiles give the workgroups that keep the weight stream wide
    const bool full = g.N >= 2048 && epi != EPI_F32_STORE;      // (the fp32 epilogue exists in the half-tile form)
    const int nhalf = (g.N + 15) / 16;
    if (!full && mt == 1 && r.M > 16 && (nhalf & 7) == 0 && (r.N & 63) == 0 && epi == EPI_HEADS) {
        // one row tile, few output features (cross-Q): the second projection split by token group, 2*nhalf units; with
        // nres a multiple of 8 the two units of a slice keep the same XCD
        const int nres_s = r.N / 8;
        const size_t shs = (size_t)8 * 8 * 64 * sizeof(float) + 32 * sizeof(float);
        MG_LAUNCH((gemm_rows_pair_split_kernel<EPI_HEADS>), dim3(nres_s + 2 * nhalf), dim3(512), shs, stream, r, g, nres_s);
        return;
    }
    const bool f16 = mt >= 2 && (r.N & 15) == 0 && g_resid_f16;      // residual part: 16 features per workgroup (resid_block16 F16)
    const bool ft2 = !full && mt >= 3 && rows_ft2(g) && (g.N % 32) == 0 && epi == EPI_HEADS && f16;      // (rows_block16 FT = 2, as gemm_rows: same bits)
    const int nres = f16 ? r.N / 16 : r.N / 8, nrows = ((g.N + 31) / 32) * ((full || ft2) ? 1 : 2);
    const dim3 grid(nres + nrows), block(512);
    const size_t sh = (size_t)8 * 16 * 64 * sizeof(float) + (size_t)32 * mt * sizeof(float);
    if (ft2) {
#define MG_RPF(MTV) case MTV: MG_LAUNCH((gemm_rows_pair_kernel<EPI_HEADS, MTV, true, true, 2>), grid, block, sh, stream, r, g, nres); return;
        switch (mt) { MG_RPF(3) MG_RPF(4) MG_RPF(5) MG_RPF(6) MG_RPF(7) MG_RPF(8) default: break; }
#undef MG_RPF
    }
#define MG_RP2(MTV, FV)                                                                                              \
        if (epi == EPI_PK_RELU && full) MG_LAUNCH((gemm_rows_pair_kernel<EPI_PK_RELU, MTV, false, FV>), grid, block, sh, stream, r, g, nres); \
        else if (epi == EPI_PK_RELU) MG_LAUNCH((gemm_rows_pair_kernel<EPI_PK_RELU, MTV, true, FV>), grid, block, sh, stream, r, g, nres); \
        else if (epi == EPI_F32_STORE) MG_LAUNCH((gemm_rows_pair_kernel<EPI_F32_STORE, MTV, true, FV>), grid, block, sh, stream, r, g, nres); \
        else MG_LAUNCH((gemm_rows_pair_kernel<EPI_HEADS, MTV, true, FV>), grid, block, sh, stream, r, g, nres);
#define MG_RP(MTV)                                                                                                   \
    case MTV:                                                                                                        \
        if (f16) { MG_RP2(MTV, true) } else { MG_RP2(MTV, false) }                                                   \
        break;
    switch (mt) {
        MG_RP(1) MG_RP(2) MG_RP(3) MG_RP(4) MG_RP(5) MG_RP(6) MG_RP(7) MG_RP(8)
        default: break;
    }
#undef MG_RP2
#undef MG_RP
}

// fp32 helpers for product weights (finalize time, not on the hot path)
__global__ __launch_bounds__(256) void unpack_weight_kernel(const uint16_t* W, float* out, int N, int K) {
    const size_t n_el = (size_t)N * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / K), k = (int)(i - (size_t)n * K);
        out[i] = bf16_to_f32(W[pk_off(n, k, K)]);
    }
}
void unpack_weight(const uint16_t* W_pk, float* out, int N, int K, mgStream_t stream) {
    MG_LAUNCH(unpack_weight_kernel, dim3(1024), dim3(256), 0, stream, W_pk, out, N, K);
}
// 32x32 output tile per workgroup, 16x16 threads with 2x2 outputs each, k staged through LDS in chunks of 32
__global__ __launch_bounds__(256) void gemm_f32_scaled_kernel(const float* A, const float* gain, const float* B, float* C, int N, int K,
                                                         int J, int ldc) {
    MG_DYN_SMEM(smem);
    float* As = (float*)smem;            // [32][33]
    float* Bs = As + 32 * 33;            // [32][33]
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int tiles_j = (J + 31) / 32;
    const int n0 = (blockIdx.x / tiles_j) * 32, j0 = (blockIdx.x % tiles_j) * 32;
    float c00 = 0.f, c01 = 0.f, c10 = 0.f, c11 = 0.f;
    for (int k0 = 0; k0 < K; k0 += 32) {
        for (int e = threadIdx.x; e < 1024; e += 256) {
            const int r = e >> 5, c = e & 31;
            const int n = n0 + r, k = k0 + c;
            As[r * 33 + c] = (n < N && k < K) ? A[(size_t)n * K + k] * gain[k] : 0.f;
            const int kb = k0 + r, j = j0 + c;
            Bs[r * 33 + c] = (kb < K && j < J) ? B[(size_t)kb * J + j] : 0.f;
        }
        __syncthreads();
        for (int kk = 0; kk < 32; ++kk) {
            const float a0 = As[(2 * ty) * 33 + kk], a1 = As[(2 * ty + 1) * 33 + kk];
            const float b0 = Bs[kk * 33 + 2 * tx], b1 = Bs[kk * 33 + 2 * tx + 1];
            c00 += a0 * b0; c01 += a0 * b1; c10 += a1 * b0; c11 += a1 * b1;
        }
        __syncthreads();
    }
    const int n = n0 + 2 * ty, j = j0 + 2 * tx;
    if (n < N && j < J) C[(size_t)n * ldc + j] = c00;
    if (n < N && j + 1 < J) C[(size_t)n * ldc + j + 1] = c01;
    if (n + 1 < N && j < J) C[(size_t)(n + 1) * ldc + j] = c10;
    if (n + 1 < N && j + 1 < J) C[(size_t)(n + 1) * ldc + j + 1] = c11;
}
void gemm_f32_scaled(const float* A, const float* gain, const float* B, float* C, int N, int K, int J, int ldc, mgStream_t stream) {
    const int blocks = ((N + 31) / 32) * ((J + 31) / 32);
    MG_LAUNCH(gemm_f32_scaled_kernel, dim3(blocks), dim3(256), 2 * 32 * 33 * sizeof(float), stream, A, gain, B, C, N, K, J, ldc);
}
__global__ __launch_bounds__(256) void scale_cols_f32_kernel(const float* A, const float* gain, float* C, int N, int J, int ldc) {
    const size_t n_el = (size_t)N * J;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / J), j = (int)(i - (size_t)n * J);
        C[(size_t)n * ldc + j] = A[i] * gain[j];
    }
}
void scale_cols_f32(const float* A, const float* gain, float* C, int N, int J, int ldc, mgStream_t stream) {
    MG_LAUNCH(scale_cols_f32_kernel, dim3(1024), dim3(256), 0, stream, A, gain, C, N, J, ldc);
}

int splitk_factor(int N, int K) {
    const int ntiles = (N + 31) / 32;
    int ks = (320 + ntiles - 1) / ntiles;       // aim for >= ~1.25 workgroups per CU
    const int ks_max = K / 64;                   // every wave keeps at least one 16-wide k-tile
    if (ks > ks_max) ks = ks_max;
    if (ks > 16) ks = 16;
    if (ks < 1) ks = 1;
    return ks;
}

void gemm_rows_splitk(const uint16_t* X, const uint16_t* W, float* P, int M, int N, int K, int ldp, size_t slab_stride, int KS,
                      const RowScale& rs, mgStream_t stream, const TopOut* top_in) {
    TopOut top{};
    if (top_in && KS == 1) top = *top_in;
    const int mt = (M + 31) / 32;
    const dim3 grid(((N + 31) / 32) * KS), block(256);
    const size_t sh = (size_t)4 * 16 * 64 * sizeof(float) + (size_t)32 * mt * sizeof(float) + (top.ptop ? (size_t)4 * 8 * 33 * sizeof(float) : 0);
    switch (mt) {
        case 1: MG_LAUNCH((gemm_rows_splitk_kernel<1>), grid, block, sh, stream, X, W, P, M, N, K, ldp, slab_stride, KS, rs, top); break;
        case 2: MG_LAUNCH((gemm_rows_splitk_kernel<2>), grid, block, sh, stream, X, W, P, M, N, K, ldp, slab_stride, KS, rs, top); break;
        case 3: MG_LAUNCH((gemm_rows_splitk_kernel<3>), grid, block, sh, stream, X, W, P, M, N, K, ldp, slab_stride, KS, rs, top); break;
        case 4: MG_LAUNCH((gemm_rows_splitk_kernel<4>), grid, block, sh, stream, X, W, P, M, N, K, ldp, slab_stride, KS, rs, top); break;
        case 5: MG_LAUNCH((gemm_rows_splitk_kernel<5>), grid, block, sh, stream, X, W, P, M, N, K, ldp, slab_stride, KS, rs, top); break;
        case 6: MG_LAUNCH((gemm_rows_splitk_kernel<6>), grid, block, sh, stream, X, W, P, M, N, K, ldp, slab_stride, KS, rs, top); break;
        case 7: MG_LAUNCH((gemm_rows_splitk_kernel<7>), grid, block, sh, stream, X, W, P, M, N, K, ldp, slab_stride, KS, rs, top); break;
        case 8: MG_LAUNCH((gemm_rows_splitk_kernel<8>), grid, block, sh, stream, X, W, P, M, N, K, ldp, slab_stride, KS, rs, top); break;
        default: break;
    }
}

void gemm_rows(const GemmArgs& a_in, int epi, mgStream_t stream) {
#ifdef MG_TOOLS      // what-if variants with WRONG results (tools build only): MG_WHATIF_KV = 1: the per-head projection does not append K / V to the
    GemmArgs a = a_in;   // cache; 2: it appends them at position 0 whatever the step
    {
        static int wi = -1;
        if (wi < 0) { const char* e = getenv("MG_WHATIF_KV"); wi = e ? atoi(e) : 0; }
        if (wi && epi == EPI_HEADS) {
            for (int ri = 0; ri < 3; ++ri)
                if (a.heads.fmt[ri] == HF_STEP_KV) {
                    if (wi == 1) a.heads.fmt[ri] = HF_NONE;
                    else { a.heads.pos_rows = nullptr; a.heads.pos_dev = nullptr; a.heads.pos = 0; }
                }
        }
    }
#else
    const GemmArgs& a = a_in;
#endif
    const int mt = (a.M + 31) / 32;
    if (mt > 8) {   // many live rows: the tiled kernel is the better shape
        gemm(a, epi, stream);
        return;
    }
    // projections with few feature tiles get one workgroup per 16 features (packed / per-head epilogues only)
    const bool half = (epi == EPI_PK_RELU || epi == EPI_PK || epi == EPI_HEADS) && ((a.N + 31) / 32) < 256 && (a.N % 16) == 0;
    switch (epi) {
        case EPI_F32_STORE:      // few feature tiles, no bias, 16-byte aligned rows: the half-tile form (twice the workgroups, 8 waves splitting K)
            gemm_rows_mt<EPI_F32_STORE>(a, mt, !a.bias && ((a.N + 31) / 32) < 128 && (a.N % 16) == 0 && (a.ldo % 4) == 0, stream); break;
        case EPI_F32_RESID: gemm_rows_mt<EPI_F32_RESID>(a, mt, false, stream); break;
        case EPI_PK_RELU: gemm_rows_mt<EPI_PK_RELU>(a, mt, half, stream); break;
        case EPI_PK: gemm_rows_mt<EPI_PK>(a, mt, half, stream); break;
        case EPI_PK_SWIGLU: gemm_rows_mt_half<EPI_PK_SWIGLU>(a, mt, stream); break;      // N % 16 == 0
        default: gemm_rows_mt<EPI_HEADS>(a, mt, half, stream); break;
    }
}

}  // namespace mg
